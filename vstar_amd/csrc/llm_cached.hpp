// llm_cached.hpp — the KV-cached LLaMA runner shared by the two engines (dtype-generic, VS_NS): the VQA-LLM's forward
// (LLaVA/llava/model/language_model/llava_search_llama.py:56-113) and the VSM's free-text decode for the contextual-cue
// branch (VSMForCausalLM.inference mode='vqa', VisualSearch/model/VSM.py:438-462 / visual_search.py:427-443).
// It owns the KV cache, the activations and the row metadata; the weights belong to the engine that built them.
//
// One forward call advances nseq sequences by any number of new rows each, in one of two regimes:
//   * prefill  (every past_len == 0 and more than 64 new rows): sequences right-padded to a common length, the MFMA tile
//     GEMMs and the flash-attention kernel; K/V rows are stored into the cache on the way.
//   * cached   (decode steps, option continuations): flat ragged rows, weight-streaming skinny GEMMs (M <= 64), attention
//     straight out of the KV cache with an optional shared prefix slot.
#pragma once
#include "engine_base.hpp"

namespace VS_NS {

struct LlmCachedCfg {
  int hidden = 0, heads = 0, mlp = 0, layers = 0, vocab = 0;
  float rms_eps = 1e-6f, rope_theta = 10000.f;
  int max_slots = 1, max_ctx = 1024, max_rows = 1024;
};

struct LlmCached {
  EngineBase* e = nullptr;
  LlmCachedCfg cfg;
  bool ready = false;
  // weights (not owned)
  const lp_t* embed = nullptr;
  const std::vector<LlmBlock>* blocks = nullptr;
  const lp_t* final_norm = nullptr;
  const Lin* lm_head = nullptr;
  // rows < 0 of `src` index this table (device, [n_feat_rows, hidden]); set by the owner before forward()
  const lp_t* feats = nullptr;
  int64_t n_feat_rows = 0;
  // owned
  lp_t* rope = nullptr;                          // [max_ctx, 128] cos | sin
  lp_t *kcache = nullptr, *vcache = nullptr;     // [layers][slots][heads][ctx][128]
  int64_t slot_stride = 0, layer_stride = 0;
  lp_t *lx = nullptr, *lh = nullptr, *lqkv = nullptr, *latt = nullptr, *lact = nullptr;
  lp_t *wsel = nullptr, *wnorm = nullptr, *logits = nullptr;
  int32_t *d_src = nullptr, *d_row_pos = nullptr, *d_row_slot = nullptr, *d_row_seq = nullptr, *d_seq = nullptr, *d_want = nullptr,
          *d_argmax = nullptr;
  int max_want = 256;
  static constexpr int SPLIT_ROWS = 4;           // decode steps of up to this many sequences take the split-KV attention
  char* split_ws = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  double last_ms = 0;
  // ---- greedy decode of ONE sequence as a replayed hipGraph (decode_greedy_graph) ----
  // A decode step is ~165 launches of 10-55 us (5 per layer); their arguments never change from token to token — the token id,
  // position and past length live in DEVICE arrays — so the step is captured once and replayed; a last kernel of the graph feeds
  // the arg-max back as the next input and advances the position, and the host looks at the emitted tokens every few steps only.
  hipGraphExec_t dec_exec = nullptr;
  int dec_keys_bound = 0;                        // context bound the captured attention launch was sized for
  int32_t* d_dec = nullptr;                      // coherent HOST memory: [0] = tokens emitted so far, [1..] = the tokens
  int dec_cap = 0;
  // ---- tile-major copies of the big decode weights (round 5; GemmParams::W_tiled) ----
  // The weight-streaming GEMV gives a workgroup 16 (gate|up: 32) ROWS of the row-major matrix, i.e. 16 - 32 sequential streams 2 K
  // bytes apart and 4096 - 22016 of them over the chip; laid out tile-major the same bytes stream at 6.5 - 7.0 TB/s instead of
  // 5.4 - 6.5 (profiles/r05_stream_layout_probe.txt).  HBM has the room for a second copy (q|k|v, gate|up, down: 11.8 GB at 7B);
  // built when the runner is initialised, VSTAR_DECODE_TILED=0 keeps the row-major streams (A/B, tests: bit-identical).
  std::vector<lp_t*> wt_qkv, wt_gate_up, wt_down;
  bool tiled_fallback = false;      // the tile-major copies were wanted but did not fit / pack: decode runs on the row-major weights

  void set_error(const std::string& m) { e->set_error(m); }
  int init(EngineBase* owner, const LlmCachedCfg& c, const lp_t* embed_, const std::vector<LlmBlock>* blocks_,
           const lp_t* final_norm_, const Lin* lm_head_);
  void release() {
    if (ev0) hipEventDestroy(ev0);
    if (ev1) hipEventDestroy(ev1);
    ev0 = ev1 = nullptr;
    if (dec_exec) hipGraphExecDestroy(dec_exec);
    dec_exec = nullptr;
    if (d_dec) hipHostFree(d_dec);
    d_dec = nullptr;
  }
  int decode_step_body(int keys_bound);
  int decode_greedy_graph(int32_t first_token, int past, int slot, int max_new, int eos_id, int32_t* out_ids, int* n_out, bool* used);
  int lin_auto(const lp_t* A, int64_t lda, const Lin& L, void* C, int64_t ldc, int M, int epi = VSTAR_EPI_NONE,
               const lp_t* res = nullptr, int64_t ldr = 0, const lp_t* Wt = nullptr);
  int lin_norm(const lp_t* x, const lp_t* norm_w, lp_t* scratch, const Lin& L, void* C, int64_t ldc, int M, int epi,
               const lp_t* Wt = nullptr);
  int llm_layers_prefill(int nseq, int S);
  int llm_layers_cached(int R, int nseq, int max_keys, bool single_rows);
  int forward(int nseq, const int32_t* row_off, const int32_t* src, const int32_t* kv_slot, const int32_t* prefix_slot,
              const int32_t* past_len, int n_want, const int32_t* want, uint16_t* logits_out, int32_t* argmax_out);
};

inline int LlmCached::init(EngineBase* owner, const LlmCachedCfg& c, const lp_t* embed_, const std::vector<LlmBlock>* blocks_,
                           const lp_t* final_norm_, const Lin* lm_head_) {
  e = owner; cfg = c; embed = embed_; blocks = blocks_; final_norm = final_norm_; lm_head = lm_head_;
  const int H = c.hidden;
  if (H != c.heads * 128) { set_error("LLaMA head dim must be 128"); return VSTAR_ERR_INVALID; }
  {  // HF LlamaRotaryEmbedding: fp32 cos/sin cast to the activation dtype before use
    std::vector<lp_t> tab((size_t)c.max_ctx * 128);
    for (int s = 0; s < c.max_ctx; ++s)
      for (int i = 0; i < 64; ++i) {
        const float inv = 1.0f / powf(c.rope_theta, (float)(2 * i) / 128.0f);
        const float f = (float)s * inv;
        tab[(size_t)s * 128 + i] = f2lp(cosf(f));
        tab[(size_t)s * 128 + 64 + i] = f2lp(sinf(f));
      }
    RC(e->dalloc(&rope, tab.size()));
    if (hipMemcpy(rope, tab.data(), tab.size() * 2, hipMemcpyHostToDevice) != hipSuccess) { set_error("rope table upload failed"); return VSTAR_ERR_HIP; }
  }
  slot_stride = (int64_t)c.heads * c.max_ctx * 128;
  layer_stride = slot_stride * c.max_slots;
  RC(e->dalloc(&kcache, (size_t)layer_stride * c.layers));
  RC(e->dalloc(&vcache, (size_t)layer_stride * c.layers));
  const size_t R = (size_t)c.max_rows;
  RC(e->dalloc(&lx, R * H));
  RC(e->dalloc(&lh, R * H));
  RC(e->dalloc(&lqkv, R * 3 * H));
  RC(e->dalloc(&latt, R * H));
  RC(e->dalloc(&lact, R * c.mlp));
  const size_t vpad = (size_t)(c.vocab + 255) / 256 * 256;
  RC(e->dalloc(&wsel, (size_t)max_want * H));
  RC(e->dalloc(&wnorm, (size_t)max_want * H));
  RC(e->dalloc(&logits, (size_t)max_want * vpad));
  RC(e->dalloc(&d_src, R));
  RC(e->dalloc(&d_row_pos, R));
  RC(e->dalloc(&d_row_slot, R));
  RC(e->dalloc(&d_row_seq, R));
  RC(e->dalloc(&d_seq, (size_t)3 * c.max_slots * 4));
  RC(e->dalloc(&d_want, (size_t)max_want));
  RC(e->dalloc(&d_argmax, (size_t)max_want));
  {  // split-KV decode attention (decode.hip): scores / partials / tickets for steps of up to SPLIT_ROWS sequences, zeroed once
    const size_t wb = cached_attention_split_ws_bytes(SPLIT_ROWS, c.heads, c.max_ctx);
    RC(e->dalloc(&split_ws, wb));
    if (hipMemset(split_ws, 0, wb) != hipSuccess) { set_error("split-KV workspace memset failed"); return VSTAR_ERR_HIP; }
  }
  if (hipEventCreate(&ev0) != hipSuccess || hipEventCreate(&ev1) != hipSuccess) { set_error("hipEventCreate failed"); return VSTAR_ERR_HIP; }
  {  // tile-major copies of q|k|v, gate|up and down for the decode GEMV (o_proj's 33 MB live in the Infinity Cache either way).
     // OPTIONAL: the row-major weights serve the same GEMV (W_tiled == nullptr), so a failed allocation or pack — ~11.8 GB more at
     // 7B — drops the copies, clears the error and continues row-major instead of failing generate().
    const char* env = getenv("VSTAR_DECODE_TILED");
    const bool on = !(env && atoi(env) == 0);
    wt_qkv.clear(); wt_gate_up.clear(); wt_down.clear();      // a retried init starts from empty lists: wt_*[i] is layer i or nothing
    std::vector<void*> mine;                                  // this block's allocations, freed together if any step fails
    bool ok = true;
    auto tile = [&](const Lin& L, int nt, std::vector<lp_t*>& out) {
      const int rows = (L.N + 16 * nt - 1) / (16 * nt) * (16 * nt);     // (the packed W is padded to 256 rows)
      if (L.N % (16 * nt) || L.K % 64) { out.push_back(nullptr); return; }
      void* t = nullptr;
      if (hipMalloc(&t, (size_t)rows * L.K * sizeof(lp_t)) != hipSuccess) { (void)hipGetLastError(); ok = false; return; }
      mine.push_back(t);
      if (skinny_pack_tiles(L.W, (lp_t*)t, rows, L.K, nt, e->stream) != hipSuccess) { (void)hipGetLastError(); ok = false; return; }
      out.push_back((lp_t*)t);
    };
    if (on && c.hidden >= 512) {
      size_t need = 0, free_b = 0, total_b = 0;
      for (int i = 0; i < c.layers; ++i) {
        const LlmBlock& b = (*blocks)[i];
        need += ((size_t)b.qkv.N * b.qkv.K + (size_t)b.gate_up.N * b.gate_up.K + (size_t)b.down.N * b.down.K) * sizeof(lp_t);
      }
      // keep 1 GiB of headroom for the caller's later allocations (KV growth happens above; this is the last big block of init)
      if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < need + ((size_t)1 << 30)) { (void)hipGetLastError(); ok = false; }
      for (int i = 0; ok && i < c.layers; ++i) {
        const LlmBlock& b = (*blocks)[i];
        tile(b.qkv, 1, wt_qkv);
        if (ok) tile(b.gate_up, 2, wt_gate_up);
        if (ok) tile(b.down, 1, wt_down);
      }
      if (hipStreamSynchronize(e->stream) != hipSuccess) { (void)hipGetLastError(); ok = false; }
      if (ok) {
        for (void* q : mine) e->allocs.push_back(q);            // owned by the engine from here on
      } else {
        for (void* q : mine) hipFree(q);
        wt_qkv.clear(); wt_gate_up.clear(); wt_down.clear();    // row-major decode (callers test .empty())
        tiled_fallback = true;
      }
    }
  }
  ready = true;
  return 0;
}

#define LCHK(expr)                                                                           \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess) {                                                                  \
      set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                          \
      return VSTAR_ERR_HIP;                                                                  \
    }                                                                                        \
  } while (0)

// GEMM dispatch for the language model: weight-streaming kernel for decode-sized M, MFMA tile kernels otherwise
inline int LlmCached::lin_auto(const lp_t* A, int64_t lda, const Lin& L, void* C, int64_t ldc, int M, int epi, const lp_t* res,
                               int64_t ldr, const lp_t* Wt) {
  GemmParams p{};
  p.A = A; p.lda = lda; p.W = L.W; p.bias = L.b; p.res = res; p.ldr = ldr; p.C = C; p.ldc = ldc; p.M = M; p.N = L.N; p.K = L.K;
  p.W_tiled = Wt;
  if (gemm_skinny_eligible(p)) {
    const hipError_t he = gemm_skinny_lp(p, epi, false, e->stream);
    if (he != hipSuccess) { set_error(std::string("skinny gemm launch: ") + hipGetErrorString(he)); return VSTAR_ERR_HIP; }
    return 0;
  }
  return e->gemm(p, epi, false);
}

// RMSNorm + Linear: for decode-sized M the norm is fused into the weight-streaming GEMM's operand load (bit-identical to
// the two-kernel form), otherwise norm kernel into `scratch`, then the GEMM
inline int LlmCached::lin_norm(const lp_t* x, const lp_t* norm_w, lp_t* scratch, const Lin& L, void* C, int64_t ldc, int M,
                               int epi, const lp_t* Wt) {
  const int H = cfg.hidden;
  GemmParams p{};
  p.A = x; p.lda = H; p.W = L.W; p.bias = L.b; p.C = C; p.ldc = ldc; p.M = M; p.N = L.N; p.K = L.K;
  p.W_tiled = Wt;
  if (M <= 16 && L.K == H && gemm_skinny_eligible(p)) {
    p.norm_w = norm_w; p.norm_eps = cfg.rms_eps;
    const hipError_t he = gemm_skinny_lp(p, epi, false, e->stream);
    if (he != hipSuccess) { set_error(std::string("skinny gemm launch: ") + hipGetErrorString(he)); return VSTAR_ERR_HIP; }
    return 0;
  }
  LCHK(rmsnorm_lp(x, norm_w, scratch, M, H, cfg.rms_eps, nullptr, e->stream));
  return lin_auto(scratch, H, L, C, ldc, M, epi, nullptr, 0, Wt);
}

inline int LlmCached::llm_layers_prefill(int nseq, int S) {
  const LlmCachedCfg& c = cfg;
  const int H = c.hidden, rows = nseq * S;
  const float att_scale = 1.0f / sqrtf(128.0f);
  for (int i = 0; i < c.layers; ++i) {
    const LlmBlock& b = (*blocks)[i];
    LCHK(rmsnorm_lp(lx, b.in_norm, lh, rows, H, c.rms_eps, nullptr, e->stream));
    RC(e->lin(lh, H, b.qkv, lqkv, 3 * H, rows));
    LCHK(rope_kv_append(lqkv, rope, d_row_pos, d_row_slot, kcache + (int64_t)i * layer_stride, vcache + (int64_t)i * layer_stride,
                        slot_stride, c.max_ctx, rows, c.heads, e->stream));
    LCHK(attn_forward(lqkv, latt, nseq, S, c.heads, 128, 1, att_scale, e->stream));
    RC(e->lin(latt, H, b.o, lx, H, rows, VSTAR_EPI_NONE, lx, H));
    LCHK(rmsnorm_lp(lx, b.post_norm, lh, rows, H, c.rms_eps, nullptr, e->stream));
    RC(e->lin(lh, H, b.gate_up, lact, c.mlp, rows, VSTAR_EPI_SILU_MUL));
    RC(e->lin(lact, c.mlp, b.down, lx, H, rows, VSTAR_EPI_NONE, lx, H));
  }
  return 0;
}

inline int LlmCached::llm_layers_cached(int R, int nseq, int max_keys, bool single_rows) {
  const LlmCachedCfg& c = cfg;
  const int H = c.hidden;
  const int32_t *d_kv = d_seq, *d_prefix = d_seq + c.max_slots * 4, *d_past = d_seq + 2 * c.max_slots * 4;
  (void)nseq;
  for (int i = 0; i < c.layers; ++i) {
    const LlmBlock& b = (*blocks)[i];
    lp_t* kc = kcache + (int64_t)i * layer_stride;
    lp_t* vc = vcache + (int64_t)i * layer_stride;
    RC(lin_norm(lx, b.in_norm, lh, b.qkv, lqkv, 3 * H, R, VSTAR_EPI_NONE, wt_qkv.empty() ? nullptr : wt_qkv[i]));
    // decode steps (one new row per sequence): RoPE + cache append happen inside the attention kernel
    if (!single_rows) LCHK(rope_kv_append(lqkv, rope, d_row_pos, d_row_slot, kc, vc, slot_stride, c.max_ctx, R, c.heads, e->stream));
    LCHK(cached_attention(lqkv, kc, vc, d_row_seq, d_row_pos, d_kv, d_prefix, d_past, single_rows ? rope : nullptr, latt, R,
                          c.heads, c.max_ctx, slot_stride, max_keys, e->stream, split_ws, SPLIT_ROWS));
    RC(lin_auto(latt, H, b.o, lx, H, R, VSTAR_EPI_NONE, lx, H));
    RC(lin_norm(lx, b.post_norm, lh, b.gate_up, lact, c.mlp, R, VSTAR_EPI_SILU_MUL, wt_gate_up.empty() ? nullptr : wt_gate_up[i]));
    RC(lin_auto(lact, c.mlp, b.down, lx, H, R, VSTAR_EPI_NONE, lx, H, wt_down.empty() ? nullptr : wt_down[i]));
  }
  return 0;
}

inline int LlmCached::forward(int nseq, const int32_t* row_off, const int32_t* src, const int32_t* kv_slot,
                              const int32_t* prefix_slot, const int32_t* past_len, int n_want, const int32_t* want,
                              uint16_t* logits_out, int32_t* argmax_out) {
  if (!ready) { e->set_error("language-model runner not initialised"); return VSTAR_ERR_STATE; }
  const LlmCachedCfg& c = cfg;
  if (nseq <= 0 || nseq > c.max_slots * 4 || !row_off || !src || !kv_slot || !prefix_slot || !past_len || n_want < 0 ||
      n_want > max_want || (n_want && !want)) {
    e->set_error("llm forward: bad argument");
    return VSTAR_ERR_INVALID;
  }
  LCHK(hipSetDevice(e->device));
  const int H = c.hidden;
  const int R = row_off[nseq];
  int maxT = 0, max_keys = 0;
  bool all_fresh = true;
  for (int i = 0; i < nseq; ++i) {
    const int T = row_off[i + 1] - row_off[i];
    if (T <= 0 || past_len[i] < 0 || past_len[i] + T > c.max_ctx) { e->set_error("sequence length exceeds max_ctx (or is empty)"); return VSTAR_ERR_INVALID; }
    if (kv_slot[i] < 0 || kv_slot[i] >= c.max_slots || prefix_slot[i] < 0 || prefix_slot[i] >= c.max_slots) {
      e->set_error("KV slot out of range");
      return VSTAR_ERR_INVALID;
    }
    if (past_len[i] == 0 && prefix_slot[i] != kv_slot[i]) { e->set_error("prefix slot without a prefix"); return VSTAR_ERR_INVALID; }
    maxT = T > maxT ? T : maxT;
    max_keys = past_len[i] + T > max_keys ? past_len[i] + T : max_keys;
    all_fresh = all_fresh && past_len[i] == 0;
  }
  for (int j = 0; j < n_want; ++j)
    if (want[j] < 0 || want[j] >= R) { e->set_error("want row out of range"); return VSTAR_ERR_INVALID; }
  const bool prefill = all_fresh && R > 64;
  const int rows = prefill ? nseq * maxT : R;
  if (rows > c.max_rows) { e->set_error("too many rows for one forward call (max_rows)"); return VSTAR_ERR_INVALID; }
  // ---- row metadata ----
  std::vector<int32_t> h_src((size_t)rows, INT32_MIN), h_pos((size_t)rows, -1), h_slot((size_t)rows, 0), h_seq((size_t)rows, 0);
  std::vector<int32_t> h_want((size_t)(n_want ? n_want : 1), 0), remap((size_t)R);
  for (int i = 0; i < nseq; ++i) {
    const int T = row_off[i + 1] - row_off[i];
    for (int t = 0; t < T; ++t) {
      const int r = prefill ? i * maxT + t : row_off[i] + t;
      h_src[r] = src[row_off[i] + t];
      h_pos[r] = past_len[i] + t;
      h_slot[r] = kv_slot[i];
      h_seq[r] = i;
      remap[row_off[i] + t] = r;
    }
  }
  for (int j = 0; j < n_want; ++j) h_want[j] = remap[want[j]];
  std::vector<int32_t> h_seqmeta((size_t)3 * c.max_slots * 4, 0);
  for (int i = 0; i < nseq; ++i) {
    h_seqmeta[i] = kv_slot[i];
    h_seqmeta[(size_t)c.max_slots * 4 + i] = prefix_slot[i];
    h_seqmeta[(size_t)2 * c.max_slots * 4 + i] = past_len[i];
  }
  LCHK(hipMemcpyAsync(d_src, h_src.data(), (size_t)rows * 4, hipMemcpyHostToDevice, e->stream));
  LCHK(hipMemcpyAsync(d_row_pos, h_pos.data(), (size_t)rows * 4, hipMemcpyHostToDevice, e->stream));
  LCHK(hipMemcpyAsync(d_row_slot, h_slot.data(), (size_t)rows * 4, hipMemcpyHostToDevice, e->stream));
  LCHK(hipMemcpyAsync(d_row_seq, h_seq.data(), (size_t)rows * 4, hipMemcpyHostToDevice, e->stream));
  LCHK(hipMemcpyAsync(d_seq, h_seqmeta.data(), h_seqmeta.size() * 4, hipMemcpyHostToDevice, e->stream));
  if (n_want) LCHK(hipMemcpyAsync(d_want, h_want.data(), (size_t)n_want * 4, hipMemcpyHostToDevice, e->stream));
  LCHK(hipStreamSynchronize(e->stream));       // the host vectors above go out of scope at return; keep it simple
  LCHK(hipEventRecord(ev0, e->stream));
  // ---- inputs_embeds (prepare_inputs_labels_for_multimodal, llava_search_arch.py:96-266) ----
  LCHK(embed_rows(d_src, embed, c.vocab, feats, n_feat_rows, lx, rows, H, e->stream));
  if (prefill) RC(llm_layers_prefill(nseq, maxT));
  else RC(llm_layers_cached(rows, nseq, max_keys, maxT == 1));
  // ---- model.norm + lm_head on the wanted rows (llava_search_llama.py:92-93) ----
  const size_t vpad = (size_t)(c.vocab + 255) / 256 * 256;
  if (n_want) {
    LCHK(gather_rows(lx, d_want, wsel, n_want, H, e->stream));
    RC(lin_norm(wsel, final_norm, wnorm, *lm_head, logits, (int64_t)vpad, n_want, VSTAR_EPI_NONE));
    LCHK(argmax_rows_lp(logits, n_want, c.vocab, (int64_t)vpad, d_argmax, e->stream));
  }
  LCHK(hipEventRecord(ev1, e->stream));
  if (n_want && logits_out)
    LCHK(hipMemcpy2DAsync(logits_out, (size_t)c.vocab * 2, logits, vpad * 2, (size_t)c.vocab * 2, n_want,
                            hipMemcpyDeviceToHost, e->stream));
  if (n_want && argmax_out) LCHK(hipMemcpyAsync(argmax_out, d_argmax, (size_t)n_want * 4, hipMemcpyDeviceToHost, e->stream));
  LCHK(hipStreamSynchronize(e->stream));
  float ms = 0;
  if (hipEventElapsedTime(&ms, ev0, ev1) == hipSuccess) last_ms = ms;
  e->collect_profile();
  return 0;
}

namespace {
// last node of the decode graph: the arg-max becomes the next input token, position and past length advance, the token is logged.
// `log` is coherent HOST memory (hipHostMallocCoherent): the host polls log[0] instead of synchronising the stream, so the next
// step's graph is already queued while it looks at this step's token.  Token first, counter last, system-scope release.
__global__ void decode_advance_kernel(const int32_t* __restrict__ argmax, int32_t* src, int32_t* row_pos, int32_t* seq_past, int32_t* log,
                                      int cap) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const int32_t t = argmax[0];
  src[0] = t;
  row_pos[0] += 1;
  seq_past[0] += 1;
  const int32_t n = log[0];
  if (n < cap) log[1 + n] = t;
  __threadfence_system();
  __hip_atomic_store(&log[0], n + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
}  // namespace

// one decode step of ONE sequence (row 0) with every per-step value read from device memory: what the graph captures
inline int LlmCached::decode_step_body(int keys_bound) {
  const LlmCachedCfg& c = cfg;
  const int H = c.hidden;
  LCHK(embed_rows(d_src, embed, c.vocab, feats, n_feat_rows, lx, 1, H, e->stream));
  RC(llm_layers_cached(1, 1, keys_bound, true));
  const size_t vpad = (size_t)(c.vocab + 255) / 256 * 256;
  LCHK(gather_rows(lx, d_want, wsel, 1, H, e->stream));
  RC(lin_norm(wsel, final_norm, wnorm, *lm_head, logits, (int64_t)vpad, 1, VSTAR_EPI_NONE));
  LCHK(argmax_rows_lp(logits, 1, c.vocab, (int64_t)vpad, d_argmax, e->stream));
  hipLaunchKernelGGL(decode_advance_kernel, dim3(1), dim3(64), 0, e->stream, d_argmax, d_src, d_row_pos,
                     d_seq + 2 * c.max_slots * 4, d_dec, dec_cap);
  LCHK(hipGetLastError());
  return 0;
}

// Greedy continuation of the sequence in `slot` whose cache holds `past` positions, starting from `first_token` (already the
// arg-max of the prefill): emits up to max_new - 1 further tokens, stopping at eos_id.  *used = false when the graph path is
// not available (event profiling on, a launch of the warm-up step failed): the caller then runs the stepwise loop — same kernels,
// same tokens (re-running a step rewrites the same K/V rows of the cache, so a half-finished warm-up step does no harm).
// Round 4: nothing is launched when the prefill's arg-max already is EOS; a failing warm-up step or a refused capture
// degrades to the stepwise path instead of failing generate(); the emitted tokens are read from coherent host memory with ONE
// step in flight ahead, so at most one decode step (not up to seven) runs past EOS and the stream is never drained between steps.
inline int LlmCached::decode_greedy_graph(int32_t first_token, int past, int slot, int max_new, int eos_id, int32_t* out_ids,
                                          int* n_out, bool* used) {
  *used = false;
  const LlmCachedCfg& c = cfg;
  static const bool disabled = [] { const char* v = getenv("VSTAR_DECODE_GRAPH"); return v && atoi(v) == 0; }();
  if (disabled || e->profile || max_new < 1) return 0;
  if (first_token == eos_id || max_new < 2) {          // the answer is the prefill's arg-max alone: no decode step at all
    out_ids[0] = first_token;
    *n_out = 1;
    *used = true;
    return 0;
  }
  LCHK(hipSetDevice(e->device));
  if (!d_dec) {
    dec_cap = c.max_ctx;
    void* hp = nullptr;
    if (hipHostMalloc(&hp, ((size_t)dec_cap + 1) * 4, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) {
      (void)hipGetLastError();
      return 0;                                         // no pinned memory: the stepwise loop
    }
    d_dec = (int32_t*)hp;
  }
  volatile int32_t* log = d_dec;
  const int keys_bound = c.max_ctx;                  // sizes the attention launch's LDS (4 B per key): fixed for the graph's lifetime
  // ---- per-call state of row 0 ----
  std::vector<int32_t> seqmeta((size_t)3 * c.max_slots * 4, 0);
  seqmeta[0] = slot; seqmeta[(size_t)c.max_slots * 4] = slot; seqmeta[(size_t)2 * c.max_slots * 4] = past;
  const int32_t zero = 0, pos0 = past, slot0 = slot, tok0 = first_token;
  LCHK(hipMemcpyAsync(d_src, &tok0, 4, hipMemcpyHostToDevice, e->stream));
  LCHK(hipMemcpyAsync(d_row_pos, &pos0, 4, hipMemcpyHostToDevice, e->stream));
  LCHK(hipMemcpyAsync(d_row_slot, &slot0, 4, hipMemcpyHostToDevice, e->stream));
  LCHK(hipMemcpyAsync(d_row_seq, &zero, 4, hipMemcpyHostToDevice, e->stream));
  LCHK(hipMemcpyAsync(d_seq, seqmeta.data(), seqmeta.size() * 4, hipMemcpyHostToDevice, e->stream));
  LCHK(hipMemcpyAsync(d_want, &zero, 4, hipMemcpyHostToDevice, e->stream));
  LCHK(hipStreamSynchronize(e->stream));
  d_dec[0] = 0;
  __atomic_thread_fence(__ATOMIC_SEQ_CST);
  int launched = 0;
  if (!dec_exec || dec_keys_bound != keys_bound) {
    if (dec_exec) { hipGraphExecDestroy(dec_exec); dec_exec = nullptr; }
    // the first step runs UNCAPTURED: it is a real decode step (its token is consumed below) and it takes every one-time
    // hipFuncSetAttribute of the launch helpers out of the capture
    if (decode_step_body(keys_bound) != 0 || hipStreamSynchronize(e->stream) != hipSuccess) {
      (void)hipGetLastError();
      e->set_error("");                                // not an error of generate(): the stepwise loop takes over
      return 0;
    }
    launched = 1;
    hipGraph_t graph = nullptr;
    if (hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
      const int rc = decode_step_body(keys_bound);
      const hipError_t ce = hipStreamEndCapture(e->stream, &graph);
      if (rc != 0 || ce != hipSuccess || !graph) {
        if (graph) hipGraphDestroy(graph);
        (void)hipGetLastError();
        e->set_error("");
      } else {
        const hipError_t ie = hipGraphInstantiate(&dec_exec, graph, nullptr, nullptr, 0);
        hipGraphDestroy(graph);
        if (ie != hipSuccess) { dec_exec = nullptr; (void)hipGetLastError(); }
        else dec_keys_bound = keys_bound;
      }
    } else {
      (void)hipGetLastError();                          // capture refused: the loop below launches the same step kernel by kernel
    }
  }
  *used = true;                                         // from here on this function owns the sequence's state
  int n = 0;
  out_ids[n++] = first_token;
  const int DEPTH = 2;                                  // decode steps queued ahead of the token the host is waiting for
  bool done = false;
  LCHK(hipEventRecord(ev0, e->stream));
  const int launched0 = launched;
  while (!done && n < max_new) {
    // out_ids[k] (k >= 1) = log[k], produced by decode step k; keep steps n .. n + DEPTH - 1 queued
    while (launched < std::min(max_new - 1, n - 1 + DEPTH)) {
      if (dec_exec) LCHK(hipGraphLaunch(dec_exec, e->stream));
      else RC(decode_step_body(keys_bound));
      ++launched;
    }
    if (launched < n) break;                            // max_new reached
    unsigned spins = 0;
    while (__atomic_load_n(&d_dec[0], __ATOMIC_ACQUIRE) < n) {
      if ((++spins & 4095u) == 0) {
        const hipError_t q = hipStreamQuery(e->stream);
        if (q == hipSuccess) {
          if (__atomic_load_n(&d_dec[0], __ATOMIC_ACQUIRE) >= n) break;
          set_error("decode graph: a step completed without emitting its token");
          return VSTAR_ERR_HIP;
        }
        if (q != hipErrorNotReady) { set_error(std::string("decode graph: ") + hipGetErrorString(q)); return VSTAR_ERR_HIP; }
      }
      __builtin_ia32_pause();
    }
    const int32_t t = log[n];
    out_ids[n++] = t;
    if (t == eos_id) done = true;
  }
  LCHK(hipEventRecord(ev1, e->stream));
  LCHK(hipStreamSynchronize(e->stream));
  float ms = 0;
  const int timed = launched - launched0;
  if (hipEventElapsedTime(&ms, ev0, ev1) == hipSuccess && timed > 0) last_ms = ms / timed;
  *n_out = n;
  return 0;
}

#undef LCHK

}  // namespace VS_NS
using namespace VS_NS;
