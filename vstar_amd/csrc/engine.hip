// engine.hip — host side of libvstar_hip.so: weight registry/packing, workspace, and the per-batch forward graph of
// the VSM scoring path.  Mirrors VSMForCausalLM.model_forward(inference=True) / .inference
// (VisualSearch/model/VSM.py:201-364, 438-553) with the generate() loop collapsed into one teacher-forced prefill
// (SURVEY.md §7 "hard parts"; the lm_head argmax at the verify positions lets the caller prove the collapse is exact).
#include "llm_cached.hpp"
#include "mx.hpp"
#include <atomic>
#include <tuple>
#include <cstring>

namespace {
struct SamAttn { Lin q, k, v, out; int internal; };
struct SamLayer { SamAttn self_attn, t2i, i2t; lp_t *n1g, *n1b, *n2g, *n2b, *n3g, *n3b, *n4g, *n4b; Lin lin1, lin2; };
}  // namespace

struct vstar_engine : EngineBase {
  vstar_config cfg{};
  // packed weights
  VitTower clip, owl;
  Lin projector;
  lp_t* embed = nullptr;
  std::vector<LlmBlock> llm;
  lp_t* final_norm = nullptr;
  Lin lm_head;
  lp_t* rope = nullptr;   // [Smax, 128] cos|sin bf16
  Lin det0, det1, seg0, seg1;
  lp_t *owl_post_g = nullptr, *owl_post_b = nullptr, *owl_ln_g = nullptr, *owl_ln_b = nullptr;
  Lin cls_fused, box0, box1, box2;
  Lin vis_proj;
  lp_t *no_mask = nullptr, *dense_pe = nullptr, *iou_token = nullptr, *mask_tokens = nullptr;
  SamLayer sam_layers[2];
  SamAttn sam_final; lp_t *sam_nfg = nullptr, *sam_nfb = nullptr;
  Lin conv1, conv2; lp_t *ln2d_g = nullptr, *ln2d_b = nullptr;
  Lin hyp0, hyp1, hyp2;

  // LLM activations
  int Smax = 0;
  uint8_t* lq8 = nullptr; float* lsa = nullptr;      // W8A8: quantised activation rows + per-token scales
  uint8_t* lmx = nullptr;                            // W8A8, block-scaled activations (mx.hpp): E8M0 bytes of the o_proj / down_proj inputs
  int make_lin8(const Lin& L, Lin8* out, const lp_t* fold_g = nullptr);
  uint8_t* lmxx = nullptr;                           // ... and of the residual stream (the q|k|v / gate|up inputs of the fully block-scaled chain)
  int last_w8a8_chain = 0;                           // 0 per token, 1 block-scaled o_proj / down_proj inputs, 2 + q|k|v / gate|up inputs (norms folded)
  int lin8(const uint8_t* Aq, const float* sa, const Lin& L, const Lin8& L8, void* C, int64_t ldc, int M, int epi,
           const lp_t* res, int64_t ldr, const lp_t* rope_cs = nullptr, int rope_S = 0, int rope_cols = 0, const uint8_t* a_mx = nullptr,
           uint8_t* c_mx = nullptr, uint8_t* c8 = nullptr, float* sumsq = nullptr);
  bool last_w8a8_mx = false;   // whether the last llm_forward ran o_proj / down_proj on block-scaled activations (vstar_w8a8_mx_active)
  bool fused_rope = true;      // VSTAR_FUSED_ROPE=0 keeps RoPE as a separate pass (A/B and the bit-identity test)
  // RMSNorms of the LLaMA blocks folded into the linears that consume them (default; VSTAR_FOLD_NORMS=0 before vstar_create keeps
  // the norm kernels): weight folded into W's columns at load, 1/rms applied to the accumulators, statistics from the epilogue
  // that writes the residual stream.  lr [rows] = rstd of the pass in flight, lpart [rows, H/64] = the epilogue's partial sums.
  bool fold_norms = true;
  float *lr = nullptr, *lpart = nullptr;
  lp_t *lx = nullptr, *lh = nullptr, *lqkv = nullptr, *latt = nullptr, *lact = nullptr;
  lp_t* hsel = nullptr;      // [B*(1+V), H] normed hidden rows
  lp_t *sel_att = nullptr, *sel_x = nullptr, *sel_h = nullptr, *sel_act = nullptr;   // last-block row subset
  float* vlogits = nullptr;    // [B*V, vocab]
  lp_t *fc_tmp = nullptr, *emb_det = nullptr, *emb_seg = nullptr;
  int32_t *d_ids = nullptr, *d_rowidx = nullptr, *d_argmax = nullptr;
  lp_t *d_clip_pix = nullptr, *d_owl_pix = nullptr;
  // OWL / SAM activations
  lp_t *owl_feats_pre = nullptr, *owl_feats = nullptr, *box_t0 = nullptr, *box_t1 = nullptr;
  float *cls_emb = nullptr, *box_raw = nullptr;
  lp_t *s_src = nullptr, *s_keys = nullptr, *s_kpe = nullptr, *s_ia = nullptr, *s_ib = nullptr, *s_ic = nullptr;
  lp_t *s_tok0 = nullptr, *s_q = nullptr, *s_qpe = nullptr, *s_ta = nullptr, *s_tb = nullptr, *s_tc = nullptr, *s_td = nullptr,
         *s_mlp = nullptr;
  lp_t *s_col1 = nullptr, *s_c1 = nullptr, *s_c1n = nullptr, *s_col2 = nullptr, *s_c2 = nullptr, *s_hyp_in = nullptr,
         *s_hyp_a = nullptr, *s_hyp_b = nullptr, *s_hyper = nullptr;
  int32_t* d_tokidx = nullptr;
  vstar_result* d_results = nullptr;
  int last_B = 0, last_S = 0;
  void* d_stats = nullptr;
  void* d_stats_batch = nullptr; size_t stats_batch_cap = 0;      // vstar_heatmap_stats_batch scratch
  void* comm = nullptr;                                           // ncclComm_t of vstar_comm_init (comm.hip), or null
  float* d_up = nullptr; size_t up_cap = 0;                       // vstar_upsample_mask scratch: [192*192 in | h*w out], grow-only
  // GPU-side preprocessing state
  // resident full images, one per slot (vstar_image_set_slot): crops of different images can share an engine batch
  struct ImageSlot { uint8_t* d = nullptr; size_t cap = 0; int H = 0, W = 0; };
  ImageSlot images[VSTAR_MAX_IMAGE_SLOTS];
  // asynchronous uploads (vstar_image_set_slot_async): pinned staging ring + a copy stream of their own, so that the next samples'
  // images travel to HBM while the engine stream is busy scoring; a preprocessing launch waits (on the device) for the uploads of
  // the slots it reads.  May be called from another host thread than the scoring calls.
  hipStream_t stream_up = nullptr;
  hipEvent_t ev_up[VSTAR_MAX_IMAGE_SLOTS] = {};
  std::atomic<int> up_pending[VSTAR_MAX_IMAGE_SLOTS] = {};
  uint8_t* h_stage[2] = {nullptr, nullptr};
  size_t stage_cap[2] = {0, 0};
  hipEvent_t ev_stage[2] = {nullptr, nullptr};
  bool stage_used[2] = {false, false};
  int stage_next = 0;
  hipEvent_t ev_pre = nullptr;                                     // recorded behind every preprocessing launch (engine stream)
  int image_upload_async(int slot, const uint8_t* rgb, int height, int width);
  uint8_t* d_temp = nullptr; size_t temp_cap = 0;
  int32_t* d_tables = nullptr; size_t tables_cap = 0;
  PreJob* d_jobs = nullptr;
  lp_t* d_lut = nullptr;
  std::vector<int32_t> h_tables;
  std::vector<PreJob> h_jobs;
  struct AxisTab { int off_b, off_c, ks; };
  int preprocess(int B, const int32_t* boxes, const int32_t* slots = nullptr);
  std::vector<int32_t> h_rowidx;
  // ---- the scoring step as a hipGraph (round 5, latency regime: <= 8 crops per call) ----
  // A small-batch step is ~700 launches of 5 - 80 us kernels on two streams; replaying it as ONE graph removes the per-launch host
  // work and most of the gap between consecutive kernels.  A graph is captured per launch signature — everything that changes a
  // kernel argument: crops, text length, verify positions, flags, image-token column, shared-prefix length, the pixel pointers —
  // on the SECOND call with that signature (the first runs eagerly and takes every one-time hipFuncSetAttribute out of the capture).
  // Per-call data (token ids, row indices) enters through pinned staging buffers at fixed addresses that the graph's copy nodes
  // read.  OPT-IN (VSTAR_SCORE_GRAPH=1 when the engine is created): measured on the MI355X it changes nothing — 16.50 vs 16.33 ms at
  // one crop per step, 37.6 vs 37.6 at four (profiles/r05_small_batch_graph_ab.txt): the small-batch step is bound by what its
  // kernels do on the GPU (under-filled grids at the per-CU operand-pull ceiling), not by launching them.  Event profiling and
  // host-pointer pixels bypass it.
  struct ScoreSig {
    int B, L, nv, img_col, psh; unsigned flags; const void *cpix, *opix;
    bool operator<(const ScoreSig& o) const {
      return std::tie(B, L, nv, img_col, psh, flags, cpix, opix) < std::tie(o.B, o.L, o.nv, o.img_col, o.psh, o.flags, o.cpix, o.opix);
    }
  };
  std::map<ScoreSig, hipGraphExec_t> score_graphs;      // nullptr = signature seen once (eager warm-up done), not yet captured
  int32_t* h_ids_pin = nullptr; int32_t* h_rowidx_pin = nullptr;
  bool score_graph_on = false;          // VSTAR_SCORE_GRAPH=1 at vstar_create (opt-in: measured equal to the eager path)
  int score_graph_max_B = 8;
  int64_t score_graph_replays = 0;
  int score_body(int B, int L, int n_verify, unsigned flags, int img_col, int psh, const lp_t* cpix, const lp_t* opix, bool skip_owl,
                 const int32_t* ids_src, const int32_t* rowidx_src, size_t n_rowidx);

  LlmCached gen;                // KV-cached runner for the free-text decode (built on first use: 0.5 GiB of cache)
  lp_t* gen_feats = nullptr;    // [P, H] projected image features of the crop being decoded
  int generate(const lp_t* clip_pix, const int32_t* ids, int L, int max_new, int eos_id, unsigned flags, int32_t* out_ids,
               int32_t* n_out);
  int finalize();
  int score(int B, const lp_t* clip_pix, const lp_t* owl_pix, const int32_t* ids, int L, const int32_t* loc_pos,
            const int32_t* verify_pos, int n_verify, unsigned flags, vstar_result* out);
  // grouped scoring: G crops x T prompts that share their first Lc spliced positions (see include/vstar_hip.h)
  int score_grouped(int G, int T, const lp_t* clip_pix, const lp_t* owl_pix, const int32_t* prefix_ids, int Lp,
                    const int32_t* suffix_ids, int Ls, const int32_t* loc_in_suffix, const int32_t* verify_in_suffix, int n_verify,
                    unsigned flags, vstar_result* out);
  // shared stages of score / score_grouped
  int grp_R0 = 0, grp_Lc = 0;       // grouped-sequence geometry of the LLaMA pass in flight (0 = plain sequences)
  int psh_Lp = 0;                   // shared-prefix length of the LLaMA pass in flight (VSTAR_F_SHARE_PREFIX; 0 = off)
  int llm_forward(int nseq, int S, int nsel);
  int llm_heads(int nrec, int n_verify);
  int owl_heads_sam(const lp_t* opix, int Bimg, int nrec, int img_div);
  int owl_features(const lp_t* opix, int Bimg);                   // the part of a9-a11 that does not depend on the LLaMA path
  int owl_finish(int Bimg, int nrec, int img_div);                // the part that needs embed_det / embed_seg
  // Small batches leave most CUs idle in every kernel (under-filled grids), and the OWL-ViT tower + box head + visual projection do
  // not depend on the CLIP -> LLaMA chain: they run on a SECOND stream, concurrently with it, and join before the class logits
  // (round 3; same kernels, same results).  Off for large batches (every kernel fills the chip on its own; concurrency would only
  // thrash the caches) and under event profiling.  VSTAR_OWL_OVERLAP=0/1 forces it.
  hipStream_t stream2 = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  int owl_overlap = -1;                                            // -1: automatic (crops <= 8), 0 / 1: forced
  bool fork_owl(const lp_t* opix, int Bimg, int* rc);              // true: owl_features was enqueued on stream2
  int finish_records(int nrec, int n_verify, unsigned flags, vstar_result* out);
  int stage_pixels(int B, const lp_t* clip_pix, const lp_t* owl_pix, unsigned flags, bool skip_owl, const lp_t** cpix, const lp_t** opix);
  lp_t* grp_feats = nullptr;        // [max_batch * P, H] projected image features (grouped scoring)
  int32_t* d_src = nullptr;         // [max_batch * Smax] embed_rows sources (grouped scoring)
  std::vector<int32_t> h_src;
  int sam_attn(const SamAttn& a, const lp_t* q_in, int nq, const lp_t* k_in, const lp_t* v_in, int nk, int B,
               lp_t* pq, lp_t* pk, lp_t* pv, lp_t* att, lp_t* out, const lp_t* res);
  int make_sam_attn(const std::string& pre, SamAttn* a);
};

int vstar_engine::make_sam_attn(const std::string& pre, SamAttn* a) {
  RC(make_lin({pre + "q_proj.weight"}, {pre + "q_proj.bias"}, &a->q, 256));
  RC(make_lin({pre + "k_proj.weight"}, {pre + "k_proj.bias"}, &a->k, 256));
  RC(make_lin({pre + "v_proj.weight"}, {pre + "v_proj.bias"}, &a->v, 256));
  a->internal = a->q.N;
  RC(make_lin({pre + "out_proj.weight"}, {pre + "out_proj.bias"}, &a->out, a->internal));
  return 0;
}

// W8A8 twin of a packed Linear: quantise the packed bf16 rows on the device (per output channel)
int vstar_engine::make_lin8(const Lin& L, Lin8* out, const lp_t* fold_g) {
  const int Npad = (L.N + 255) / 256 * 256;
  if (L.K % 256) { set_error("W8A8 needs K % 256 == 0"); return VSTAR_ERR_INVALID; }
  RC(dalloc(&out->W, (size_t)Npad * L.K));
  RC(dalloc(&out->s, (size_t)Npad));
  if (fold_g) {      // fp8 of round16(W diag(g)): the consumer's RMSNorm weight folded into its columns (as the bf16 path folds it)
    lp_t* tmp = nullptr;
    HIPCHK(hipMalloc((void**)&tmp, (size_t)Npad * L.K * sizeof(lp_t)));
    hipError_t e = hipMemcpyAsync(tmp, L.W, (size_t)Npad * L.K * sizeof(lp_t), hipMemcpyDeviceToDevice, stream);
    if (e == hipSuccess) e = scale_cols_lp(tmp, fold_g, Npad, L.K, stream);
    if (e == hipSuccess) e = quantize_rows_fp8(tmp, L.K, out->W, L.K, out->s, Npad, L.K, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    hipFree(tmp);
    HIPCHK(e);
    return 0;
  }
  KCHK(quantize_rows_fp8(L.W, L.K, out->W, L.K, out->s, Npad, L.K, stream));
  return 0;
}

int vstar_engine::lin8(const uint8_t* Aq, const float* sa, const Lin& L, const Lin8& L8, void* C, int64_t ldc, int M, int epi,
                       const lp_t* res, int64_t ldr, const lp_t* rope_cs, int rope_S, int rope_cols, const uint8_t* a_mx, uint8_t* c_mx,
                       uint8_t* c8, float* sumsq) {
  GemmParams p{};
  p.A = (const lp_t*)Aq; p.lda = L.K; p.W = (const lp_t*)L8.W; p.res = res; p.ldr = ldr; p.C = C; p.ldc = ldc;
  p.M = M; p.N = L.N; p.K = L.K; p.a_scale = sa; p.w_scale = L8.s; p.a_mx = a_mx; p.c_mx = c_mx;      // (sa next to a_mx: the folded norm's 1 / rms)
  p.c8 = c8; p.ldc8 = L.N; p.sumsq_out = c8 ? sumsq : nullptr; p.sumsq_ld = L.N / 64;
  p.rope_cs = rope_cs; p.rope_S = rope_S; p.rope_cols = rope_cols; p.rope_R0 = grp_R0; p.rope_Lc = grp_Lc;
  return gemm(p, epi, false);
}

int vstar_engine::finalize() {
  if (finalized) { set_error("weights already finalized"); return VSTAR_ERR_STATE; }
  const vstar_config& c = cfg;
  const int maxB = c.max_batch;
  HIPCHK(hipSetDevice(device));
  // ---- CLIP tower: hidden_states[select_layer] => clip_layers + 1 + select_layer blocks ----
  const int clip_blocks = c.clip_layers + 1 + c.clip_select_layer;
  if (clip_blocks < 0 || clip_blocks > c.clip_layers) { set_error("bad clip_select_layer"); return VSTAR_ERR_INVALID; }
  RC(build_tower(clip, "clip.vision_model.", "pre_layrnorm", c.clip_image_size, c.clip_patch, c.clip_hidden, c.clip_heads,
                 c.clip_mlp, clip_blocks, maxB));
  RC(make_lin({"model.mm_projector.weight"}, {"model.mm_projector.bias"}, &projector, c.clip_hidden));
  // ---- LLaMA ----
  const int H = c.llm_hidden;
  if (H != c.llm_heads * 128) { set_error("LLaMA head dim must be 128"); return VSTAR_ERR_INVALID; }
  if (c.llm_mlp % 16) { set_error("llm_mlp must be a multiple of 16"); return VSTAR_ERR_INVALID; }
  RC(upload_vec("model.embed_tokens.weight", &embed, (int64_t)c.llm_vocab * H));
  std::vector<int> perm(2 * c.llm_mlp);
  for (int r = 0; r < 2 * c.llm_mlp; ++r) {
    const int blk = r / 32, w = r % 32;
    perm[r] = w < 16 ? blk * 16 + w : c.llm_mlp + blk * 16 + (w - 16);
  }
  llm.resize(c.llm_layers);
  for (int i = 0; i < c.llm_layers; ++i) {
    const std::string lp = "model.layers." + std::to_string(i) + ".";
    LlmBlock& b = llm[i];
    RC(upload_vec(lp + "input_layernorm.weight", &b.in_norm, H));
    RC(upload_vec(lp + "post_attention_layernorm.weight", &b.post_norm, H));
    RC(make_lin({lp + "self_attn.q_proj.weight", lp + "self_attn.k_proj.weight", lp + "self_attn.v_proj.weight"}, {}, &b.qkv, H));
    RC(make_lin({lp + "self_attn.o_proj.weight"}, {}, &b.o, H));
    RC(make_lin({lp + "mlp.gate_proj.weight", lp + "mlp.up_proj.weight"}, {}, &b.gate_up, H, &perm));
    RC(make_lin({lp + "mlp.down_proj.weight"}, {}, &b.down, c.llm_mlp));
  }
  if (c.llm_w8a8) fold_norms = false;      // W8A8 keeps its quantisation scheme (normalised activations per token, W per channel)
  if (fold_norms) {
    // Linear(RMSNorm(x)) = rstd(x) * (x . (W * diag(norm_w))^T): the norm weights move into the columns of the q|k|v and gate|up
    // matrices (one bf16 rounding of W * w in place of the reference's two activation roundings), and the vectors become 1 so that
    // every other user of these blocks (decode runner, gathered last-block rows) stays correct as written
    if (H % 64) { set_error("fold_norms needs llm_hidden % 64 == 0"); return VSTAR_ERR_INVALID; }
    for (auto& b : llm) {
      KCHK(scale_cols_lp(b.qkv.W, b.in_norm, (int64_t)((b.qkv.N + 255) / 256 * 256), b.qkv.K, stream));
      KCHK(scale_cols_lp(b.gate_up.W, b.post_norm, (int64_t)((b.gate_up.N + 255) / 256 * 256), b.gate_up.K, stream));
      KCHK(fill_lp(b.in_norm, H, 1.0f, stream));
      KCHK(fill_lp(b.post_norm, H, 1.0f, stream));
    }
    HIPCHK(hipStreamSynchronize(stream));
  }
  if (c.llm_w8a8) {
    for (auto& b : llm) {
      RC(make_lin8(b.qkv, &b.qkv8));
      RC(make_lin8(b.o, &b.o8));
      RC(make_lin8(b.gate_up, &b.gate_up8));
      RC(make_lin8(b.down, &b.down8));
      RC(make_lin8(b.qkv, &b.qkv8f, b.in_norm));
      RC(make_lin8(b.gate_up, &b.gate_up8f, b.post_norm));
    }
  }
  RC(upload_vec("model.norm.weight", &final_norm, H));
  RC(make_lin({"lm_head.weight"}, {}, &lm_head, H));
  Smax = c.max_text_len - 1 + clip.P;
  {  // rotate-half RoPE table, HF LlamaRotaryEmbedding: inv_freq = theta^(-2i/d), fp32, cast to bf16 before use
    const int rope_rows = Smax + 160;     // grouped sequences index up to round_up(Lc, 128) and Lc + 32
    std::vector<lp_t> tab((size_t)rope_rows * 128);
    for (int s = 0; s < rope_rows; ++s)
      for (int i = 0; i < 64; ++i) {
        const float inv = 1.0f / powf(c.llm_rope_theta, (float)(2 * i) / 128.0f);
        const float f = (float)s * inv;
        tab[(size_t)s * 128 + i] = f2lp(cosf(f));
        tab[(size_t)s * 128 + 64 + i] = f2lp(sinf(f));
      }
    RC(dalloc(&rope, tab.size()));
    HIPCHK(hipMemcpy(rope, tab.data(), tab.size() * 2, hipMemcpyHostToDevice));
  }
  const size_t lrows = (size_t)(maxB + 1) * Smax;      // + 1: the shared-prefix sequence (VSTAR_F_SHARE_PREFIX)
  RC(dalloc(&lx, lrows * H));
  HIPCHK(hipMemset(lx, 0, lrows * H * sizeof(lp_t)));     // rows no call has written yet are still read by the row-wise kernels (shared-prefix slot)
  RC(dalloc(&lh, lrows * H));
  RC(dalloc(&lqkv, lrows * 3 * H));
  RC(dalloc(&latt, lrows * H));
  RC(dalloc(&lr, lrows));
  RC(dalloc(&lpart, lrows * (size_t)(H / 64 > 0 ? H / 64 : 1)));
  RC(dalloc(&lact, lrows * c.llm_mlp));
  if (c.llm_w8a8) {
    RC(dalloc(&lq8, lrows * (size_t)(c.llm_mlp > H ? c.llm_mlp : H)));
    RC(dalloc(&lsa, lrows));
    RC(dalloc(&lmx, (lrows + 127) / 128 * 128 * (size_t)((c.llm_mlp > H ? c.llm_mlp : H) / 32 + 1)));
    RC(dalloc(&lmxx, (lrows + 127) / 128 * 128 * (size_t)(H / 32 + 1)));
  }
  RC(dalloc(&hsel, (size_t)maxB * (1 + VSTAR_MAX_VERIFY) * H));
  RC(dalloc(&sel_att, (size_t)maxB * (1 + VSTAR_MAX_VERIFY) * H));
  RC(dalloc(&sel_x, (size_t)maxB * (1 + VSTAR_MAX_VERIFY) * H));
  RC(dalloc(&sel_h, (size_t)maxB * (1 + VSTAR_MAX_VERIFY) * H));
  RC(dalloc(&sel_act, (size_t)maxB * (1 + VSTAR_MAX_VERIFY) * c.llm_mlp));
  RC(dalloc(&vlogits, (size_t)maxB * VSTAR_MAX_VERIFY * c.llm_vocab));
  RC(dalloc(&d_ids, (size_t)maxB * c.max_text_len));
  RC(dalloc(&d_rowidx, (size_t)maxB * (1 + VSTAR_MAX_VERIFY)));
  // pinned staging of the per-call token ids / row indices for the scoring-step graphs (fixed addresses the copy nodes read)
  if (hipHostMalloc((void**)&h_ids_pin, (size_t)maxB * c.max_text_len * 4, hipHostMallocDefault) != hipSuccess ||
      hipHostMalloc((void**)&h_rowidx_pin, (size_t)maxB * (1 + VSTAR_MAX_VERIFY) * 4, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    h_ids_pin = h_rowidx_pin = nullptr;                 // no pinned memory: the eager path only
  }
  RC(dalloc(&d_argmax, (size_t)maxB * VSTAR_MAX_VERIFY));
  RC(dalloc(&d_clip_pix, (size_t)maxB * 3 * c.clip_image_size * c.clip_image_size));
  RC(dalloc(&d_owl_pix, (size_t)maxB * 3 * c.owl_image_size * c.owl_image_size));
  RC(dalloc(&d_results, (size_t)maxB));
  RC(dalloc(&grp_feats, (size_t)maxB * clip.P * H));
  RC(dalloc(&d_src, lrows));
  HIPCHK(hipMemset(d_results, 0, sizeof(vstar_result) * maxB));
  // ---- text_hidden_fcs ----
  RC(make_lin({"model.text_hidden_fcs_det.0.0.weight"}, {"model.text_hidden_fcs_det.0.0.bias"}, &det0, H));
  RC(make_lin({"model.text_hidden_fcs_det.0.2.weight"}, {"model.text_hidden_fcs_det.0.2.bias"}, &det1, H));
  RC(make_lin({"model.text_hidden_fcs_seg.0.0.weight"}, {"model.text_hidden_fcs_seg.0.0.bias"}, &seg0, H));
  RC(make_lin({"model.text_hidden_fcs_seg.0.2.weight"}, {"model.text_hidden_fcs_seg.0.2.bias"}, &seg1, H));
  if (det1.N != c.owl_query_dim || seg1.N != 256) { set_error("text_hidden_fcs output dims"); return VSTAR_ERR_INVALID; }
  RC(dalloc(&fc_tmp, (size_t)maxB * H));
  RC(dalloc(&emb_det, (size_t)maxB * det1.N));
  RC(dalloc(&emb_seg, (size_t)maxB * 256));
  // ---- OWL-ViT tower + heads ----
  const int OH = c.owl_hidden;
  RC(build_tower(owl, "model.owlvit.vision_model.", "pre_layernorm", c.owl_image_size, c.owl_patch, OH, c.owl_heads,
                 c.owl_mlp, c.owl_layers, maxB));
  if (owl.P != VSTAR_N_BOXES) { set_error("OWL-ViT grid must be 48x48"); return VSTAR_ERR_INVALID; }
  RC(upload_vec("model.owlvit.vision_model.post_layernorm.weight", &owl_post_g, OH));
  RC(upload_vec("model.owlvit.vision_model.post_layernorm.bias", &owl_post_b, OH));
  RC(upload_vec("model.owlvit.layer_norm.weight", &owl_ln_g, OH));
  RC(upload_vec("model.owlvit.layer_norm.bias", &owl_ln_b, OH));
  RC(make_lin({"model.owlvit.class_head.dense0.weight", "model.owlvit.class_head.logit_shift.weight",
               "model.owlvit.class_head.logit_scale.weight"},
              {"model.owlvit.class_head.dense0.bias", "model.owlvit.class_head.logit_shift.bias",
               "model.owlvit.class_head.logit_scale.bias"}, &cls_fused, OH));
  if (cls_fused.N != c.owl_query_dim + 2) { set_error("class head dims"); return VSTAR_ERR_INVALID; }
  RC(make_lin({"model.owlvit.box_head.dense0.weight"}, {"model.owlvit.box_head.dense0.bias"}, &box0, OH));
  RC(make_lin({"model.owlvit.box_head.dense1.weight"}, {"model.owlvit.box_head.dense1.bias"}, &box1, OH));
  RC(make_lin({"model.owlvit.box_head.dense2.weight"}, {"model.owlvit.box_head.dense2.bias"}, &box2, OH));
  const size_t prow = (size_t)maxB * owl.P;
  RC(dalloc(&owl_feats_pre, prow * OH));
  RC(dalloc(&owl_feats, prow * OH));
  RC(dalloc(&box_t0, prow * OH));
  RC(dalloc(&box_t1, prow * OH));
  RC(dalloc(&cls_emb, prow * (size_t)(cls_fused.N + 2)));
  RC(dalloc(&box_raw, prow * 4));
  // ---- SAM-style mask head ----
  RC(make_lin({"model.visual_projection.weight"}, {}, &vis_proj, OH));
  RC(upload_vec("model.prompt_encoder.no_mask_embed.weight", &no_mask, 256));
  RC(upload_vec("sam.dense_pe", &dense_pe, (int64_t)owl.P * 256));
  RC(upload_vec("model.mask_decoder.iou_token.weight", &iou_token, 256));
  RC(upload_vec("model.mask_decoder.mask_tokens.weight", &mask_tokens, 4 * 256));
  for (int i = 0; i < 2; ++i) {
    const std::string lp = "model.mask_decoder.transformer.layers." + std::to_string(i) + ".";
    SamLayer& L = sam_layers[i];
    RC(make_sam_attn(lp + "self_attn.", &L.self_attn));
    RC(make_sam_attn(lp + "cross_attn_token_to_image.", &L.t2i));
    RC(make_sam_attn(lp + "cross_attn_image_to_token.", &L.i2t));
    RC(upload_vec(lp + "norm1.weight", &L.n1g, 256)); RC(upload_vec(lp + "norm1.bias", &L.n1b, 256));
    RC(upload_vec(lp + "norm2.weight", &L.n2g, 256)); RC(upload_vec(lp + "norm2.bias", &L.n2b, 256));
    RC(upload_vec(lp + "norm3.weight", &L.n3g, 256)); RC(upload_vec(lp + "norm3.bias", &L.n3b, 256));
    RC(upload_vec(lp + "norm4.weight", &L.n4g, 256)); RC(upload_vec(lp + "norm4.bias", &L.n4b, 256));
    RC(make_lin({lp + "mlp.lin1.weight"}, {lp + "mlp.lin1.bias"}, &L.lin1, 256));
    RC(make_lin({lp + "mlp.lin2.weight"}, {lp + "mlp.lin2.bias"}, &L.lin2, L.lin1.N));
  }
  RC(make_sam_attn("model.mask_decoder.transformer.final_attn_token_to_image.", &sam_final));
  RC(upload_vec("model.mask_decoder.transformer.norm_final_attn.weight", &sam_nfg, 256));
  RC(upload_vec("model.mask_decoder.transformer.norm_final_attn.bias", &sam_nfb, 256));
  RC(make_lin({"model.mask_decoder.output_upscaling.0.conv.weight"}, {"model.mask_decoder.output_upscaling.0.conv.bias"},
              &conv1, 256 * 9, nullptr, true));
  RC(upload_vec("model.mask_decoder.output_upscaling.1.weight", &ln2d_g, 64));
  RC(upload_vec("model.mask_decoder.output_upscaling.1.bias", &ln2d_b, 64));
  RC(make_lin({"model.mask_decoder.output_upscaling.3.conv.weight"}, {"model.mask_decoder.output_upscaling.3.conv.bias"},
              &conv2, 64 * 9, nullptr, true));
  RC(make_lin({"model.mask_decoder.output_hypernetworks_mlps.0.layers.0.weight"},
              {"model.mask_decoder.output_hypernetworks_mlps.0.layers.0.bias"}, &hyp0, 256));
  RC(make_lin({"model.mask_decoder.output_hypernetworks_mlps.0.layers.1.weight"},
              {"model.mask_decoder.output_hypernetworks_mlps.0.layers.1.bias"}, &hyp1, 256));
  RC(make_lin({"model.mask_decoder.output_hypernetworks_mlps.0.layers.2.weight"},
              {"model.mask_decoder.output_hypernetworks_mlps.0.layers.2.bias"}, &hyp2, 256));
  if (conv1.N != 64 || conv2.N != 32 || hyp2.N != 32) { set_error("mask decoder dims"); return VSTAR_ERR_INVALID; }
  RC(dalloc(&s_src, prow * 256)); RC(dalloc(&s_keys, prow * 256)); RC(dalloc(&s_kpe, prow * 256));
  RC(dalloc(&s_ia, prow * 256)); RC(dalloc(&s_ib, prow * 256)); RC(dalloc(&s_ic, prow * 256));
  const size_t trow = (size_t)maxB * 6;
  RC(dalloc(&s_tok0, trow * 256)); RC(dalloc(&s_q, trow * 256)); RC(dalloc(&s_qpe, trow * 256));
  RC(dalloc(&s_ta, trow * 256)); RC(dalloc(&s_tb, trow * 256)); RC(dalloc(&s_tc, trow * 256)); RC(dalloc(&s_td, trow * 256));
  RC(dalloc(&s_mlp, trow * 2048));
  RC(dalloc(&s_col1, (size_t)maxB * 96 * 96 * 2304));
  RC(dalloc(&s_c1, (size_t)maxB * 96 * 96 * 64));
  RC(dalloc(&s_c1n, (size_t)maxB * 96 * 96 * 64));
  RC(dalloc(&s_col2, (size_t)maxB * 192 * 192 * 576));
  RC(dalloc(&s_c2, (size_t)maxB * 192 * 192 * 32));
  RC(dalloc(&s_hyp_in, (size_t)maxB * 256)); RC(dalloc(&s_hyp_a, (size_t)maxB * 256)); RC(dalloc(&s_hyp_b, (size_t)maxB * 256));
  RC(dalloc(&s_hyper, (size_t)maxB * 32));
  RC(dalloc(&d_tokidx, (size_t)maxB));
  {
    std::vector<int32_t> tix(maxB);
    for (int b = 0; b < maxB; ++b) tix[b] = b * 6 + 1;   // mask token 0 = token row 1 of each crop
    HIPCHK(hipMemcpy(d_tokidx, tix.data(), (size_t)maxB * 4, hipMemcpyHostToDevice));
  }
  RC(dalloc(&d_jobs, (size_t)maxB * 2));
  RC(dalloc(&d_lut, (size_t)3 * 256));
  {
    lp_t lut[3 * 256];
    clip_norm_lut(lut);
    HIPCHK(hipMemcpy(d_lut, lut, sizeof(lut), hipMemcpyHostToDevice));
  }
  HIPCHK(hipDeviceSynchronize());
  staged.clear();
  finalized = true;
  return 0;
}

// crop + pad + Pillow-exact resize + normalise for B boxes of the resident image -> d_clip_pix / d_owl_pix
int vstar_engine::preprocess(int B, const int32_t* boxes, const int32_t* slots) {
  if (!finalized) { set_error("vstar_finalize_weights has not been called"); return VSTAR_ERR_STATE; }
  if (B <= 0 || B > cfg.max_batch || !boxes) { set_error("bad preprocess arguments"); return VSTAR_ERR_INVALID; }
  for (int b = 0; b < B; ++b) {
    const int sl = slots ? slots[b] : 0;
    if (sl < 0 || sl >= VSTAR_MAX_IMAGE_SLOTS) { set_error("image slot out of range"); return VSTAR_ERR_INVALID; }
    if (!images[sl].d) { set_error("no image resident in the slot: call vstar_image_set / vstar_image_set_slot first"); return VSTAR_ERR_STATE; }
  }
  HIPCHK(hipSetDevice(device));
  HIPCHK(hipStreamSynchronize(stream));   // the staging vectors below are reused across calls
  for (int b = 0; b < B; ++b) {           // images still on their way (vstar_image_set_slot_async): the engine stream waits for them
    const int sl = slots ? slots[b] : 0;
    if (up_pending[sl].exchange(0)) HIPCHK(hipStreamWaitEvent(stream, ev_up[sl], 0));
  }
  const int I = cfg.clip_image_size, O = cfg.owl_image_size;
  h_tables.clear();
  h_jobs.assign((size_t)B * 2, PreJob{});
  std::map<std::pair<int, int>, AxisTab> cache;
  auto axis = [&](int in_size, int out_size) -> AxisTab {
    auto key = std::make_pair(in_size, out_size);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    std::vector<int32_t> bnd, cf;
    int ks = 0;
    pil_bicubic_coeffs(in_size, out_size, &bnd, &cf, &ks);
    AxisTab t{(int)h_tables.size(), 0, ks};
    h_tables.insert(h_tables.end(), bnd.begin(), bnd.end());
    t.off_c = (int)h_tables.size();
    h_tables.insert(h_tables.end(), cf.begin(), cf.end());
    cache[key] = t;
    return t;
  };
  size_t temp_bytes = 0;
  int max_h_clip = 0, max_h_owl = 0;
  for (int b = 0; b < B; ++b) {
    const int x0 = boxes[b * 4], y0 = boxes[b * 4 + 1], x1 = boxes[b * 4 + 2], y1 = boxes[b * 4 + 3];
    const int cw = x1 - x0, ch = y1 - y0;
    const ImageSlot& im = images[slots ? slots[b] : 0];
    if (x0 < 0 || y0 < 0 || cw <= 0 || ch <= 0 || x1 > im.W || y1 > im.H) { set_error("crop box outside the image"); return VSTAR_ERR_INVALID; }
    const int side = cw > ch ? cw : ch;
    for (int which = 0; which < 2; ++which) {
      PreJob& j = h_jobs[(size_t)b * 2 + which];
      j.x0 = x0; j.y0 = y0; j.cw = cw; j.ch = ch;
      j.img = im.d; j.img_w = im.W;
      j.in_w = which == 0 ? side : cw;
      j.in_h = which == 0 ? side : ch;
      j.out = which == 0 ? I : O;
      const AxisTab hx = axis(j.in_w, j.out), vy = axis(j.in_h, j.out);
      j.hb_off = hx.off_b; j.hc_off = hx.off_c; j.hks = hx.ks;
      j.vb_off = vy.off_b; j.vc_off = vy.off_c; j.vks = vy.ks;
      j.temp_off = (int64_t)temp_bytes;
      temp_bytes += ((size_t)j.in_h * j.out * 3 + 255) / 256 * 256;
      j.out_off = (int64_t)b * 3 * j.out * j.out;
      if (which == 0) max_h_clip = j.in_h > max_h_clip ? j.in_h : max_h_clip;
      else max_h_owl = j.in_h > max_h_owl ? j.in_h : max_h_owl;
    }
  }
  if (temp_bytes > temp_cap) {
    if (d_temp) HIPCHK(hipFree(d_temp));
    temp_cap = temp_bytes + temp_bytes / 4;
    HIPCHK(hipMalloc((void**)&d_temp, temp_cap));
  }
  if (h_tables.size() > tables_cap) {
    if (d_tables) HIPCHK(hipFree(d_tables));
    tables_cap = h_tables.size() * 2;
    HIPCHK(hipMalloc((void**)&d_tables, tables_cap * 4));
  }
  HIPCHK(hipMemcpyAsync(d_tables, h_tables.data(), h_tables.size() * 4, hipMemcpyHostToDevice, stream));
  HIPCHK(hipMemcpyAsync(d_jobs, h_jobs.data(), h_jobs.size() * sizeof(PreJob), hipMemcpyHostToDevice, stream));
  KCHK(preprocess_launch(d_jobs, d_tables, d_temp, d_lut, d_clip_pix, 0, B, I, max_h_clip, stream));
  KCHK(preprocess_launch(d_jobs, d_tables, d_temp, d_lut, d_owl_pix, 1, B, O, max_h_owl, stream));
  HIPCHK(hipEventRecord(ev_pre, stream));      // always: an asynchronous upload orders its DMA behind the crops already launched
  return 0;
}

// Upload of one image into `slot` WITHOUT stalling the engine stream: host copy into a pinned staging buffer (two of them: the copy
// into one overlaps the DMA out of the other), DMA on the upload stream, an event per slot that the next preprocessing of that
// slot waits for on the device.  The caller guarantees that no crop of the slot's previous image is still to be LAUNCHED (the
// stream driver recycles a slot only when its searches have ended); crops already launched are ordered in front of the DMA by
// ev_pre.  Safe to call from a second host thread while another is inside a scoring call.
int vstar_engine::image_upload_async(int slot, const uint8_t* rgb, int height, int width) {
  HIPCHK(hipSetDevice(device));
  const size_t bytes = (size_t)height * width * 3;
  // (stream_up, ev_stage and ev_pre are created in vstar_create: the prefetch thread that calls this must not create objects the
  // scoring thread reads — ADVICE r4)
  if (!ev_up[slot]) HIPCHK(hipEventCreateWithFlags(&ev_up[slot], hipEventDisableTiming));
  ImageSlot& im = images[slot];
  if (bytes > im.cap) {                     // first image of this size in the slot: allocate (grow-only)
    if (up_pending[slot].load()) HIPCHK(hipEventSynchronize(ev_up[slot]));
    if (im.d) { HIPCHK(hipStreamSynchronize(stream)); HIPCHK(hipFree(im.d)); }
    im.d = nullptr; im.cap = 0;
    HIPCHK(hipMalloc((void**)&im.d, bytes));
    im.cap = bytes;
  }
  const int k = stage_next;
  stage_next ^= 1;
  if (stage_used[k]) HIPCHK(hipEventSynchronize(ev_stage[k]));      // its previous DMA must have left the buffer
  if (bytes > stage_cap[k]) {
    if (h_stage[k]) HIPCHK(hipHostFree(h_stage[k]));
    h_stage[k] = nullptr; stage_cap[k] = 0;
    HIPCHK(hipHostMalloc((void**)&h_stage[k], bytes, hipHostMallocDefault));
    stage_cap[k] = bytes;
  }
  memcpy(h_stage[k], rgb, bytes);
  HIPCHK(hipStreamWaitEvent(stream_up, ev_pre, 0));      // (a never-recorded event is complete: the wait is a no-op before the first crop)
  HIPCHK(hipMemcpyAsync(im.d, h_stage[k], bytes, hipMemcpyHostToDevice, stream_up));
  HIPCHK(hipEventRecord(ev_stage[k], stream_up));
  stage_used[k] = true;
  HIPCHK(hipEventRecord(ev_up[slot], stream_up));
  im.H = height;
  im.W = width;
  up_pending[slot].store(1);
  return 0;
}

// segment_anything Attention.forward (transformer.py:220-242): proj -> heads -> softmax(qk/sqrt(c)) v -> out_proj (+res)
int vstar_engine::sam_attn(const SamAttn& a, const lp_t* q_in, int nq, const lp_t* k_in, const lp_t* v_in, int nk,
                           int B, lp_t* pq, lp_t* pk, lp_t* pv, lp_t* att, lp_t* out, const lp_t* res) {
  const int I = a.internal;
  RC(lin(q_in, 256, a.q, pq, I, B * nq));
  RC(lin(k_in, 256, a.k, pk, I, B * nk));
  RC(lin(v_in, 256, a.v, pv, I, B * nk));
  KCHK(small_attention(pq, pk, pv, att, B, nq, nk, 8, I / 8, stream));
  RC(lin(att, I, a.out, out, 256, B * nq, VSTAR_EPI_NONE, res, 256));
  return 0;
}

namespace {
__global__ void sam_tokens_kernel(const lp_t* iou, const lp_t* mask_tokens, const lp_t* seg, lp_t* out, int B) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)B * 6 * 256) return;
  const int c = (int)(idx % 256), t = (int)((idx / 256) % 6), b = (int)(idx / (6 * 256));
  out[idx] = t == 0 ? iou[c] : (t < 5 ? mask_tokens[(t - 1) * 256 + c] : seg[b * 256 + c]);
}
}  // namespace

// host/device pixel hand-over shared by score / score_grouped
int vstar_engine::stage_pixels(int B, const lp_t* clip_pix, const lp_t* owl_pix, unsigned flags, bool skip_owl, const lp_t** cpix,
                               const lp_t** opix) {
  const vstar_config& c = cfg;
  *cpix = clip_pix; *opix = owl_pix;
  if (flags & VSTAR_F_INTERNAL_PIXELS) {
    *cpix = d_clip_pix;   // filled by vstar_preprocess_crops on this stream
    *opix = d_owl_pix;
  } else if (!(flags & VSTAR_F_DEVICE_INPUTS)) {
    const size_t cn = (size_t)B * 3 * c.clip_image_size * c.clip_image_size;
    HIPCHK(hipMemcpyAsync(d_clip_pix, clip_pix, cn * 2, hipMemcpyHostToDevice, stream));
    *cpix = d_clip_pix;
    if (!skip_owl) {
      const size_t on = (size_t)B * 3 * c.owl_image_size * c.owl_image_size;
      HIPCHK(hipMemcpyAsync(d_owl_pix, owl_pix, on * 2, hipMemcpyHostToDevice, stream));
      *opix = d_owl_pix;
    }
  }
  return 0;
}

// ---- a5: LLaMA prefill (HF LlamaModel via llava_llama.py:93-102) over nseq sequences of S rows in lx; the last block runs on the
// nsel rows listed in d_rowidx only and leaves them in sel_x.  grp_R0 / grp_Lc describe grouped sequences (shared prefix + 32-row
// suffix blocks): they only change the RoPE positions and the attention mask — every other op is row-wise.
int vstar_engine::llm_forward(int nseq, int S, int nsel) {
  const vstar_config& c = cfg;
  const int H = c.llm_hidden;
  const int rows = nseq * S;
  const float att_scale = 1.0f / sqrtf(128.0f);
  // W8A8 (config 5): the four big linears of every block run on the fp8 MFMA when the call has enough rows for the 256^2
  // kernel; activations are quantised per token right where they are produced (inside the RMSNorm for q|k|v and gate|up,
  // by one pass over the attention output / the SiLU*up product for o_proj and down_proj)
  const bool w8 = c.llm_w8a8 && rows >= 1024;
  // Block-scaled activations (mx.hpp, round 6) for the inputs of o_proj and down_proj: quantised by their PRODUCERS (the attention
  // epilogue, the gate|up epilogue) per 32 values, scales applied inside the MFMA — no stand-alone quantisation pass.  Needs the
  // 4-wave kernel's domain (rows % 256 == 0 ...); otherwise the per-token scheme with its two passes.  VSTAR_W8A8_MX=0: A/B runs.
  static const bool mx_env = [] { const char* e = getenv("VSTAR_W8A8_MX"); return !e || atoi(e) != 0; }();
  bool mx = false;
  if (w8 && mx_env && c.llm_layers > 1) {
    GemmParams q{};
    LlmBlock& b0 = llm[0];
    q.A = (const lp_t*)latt; q.lda = H; q.W = (const lp_t*)b0.o8.W; q.C = lx; q.ldc = H; q.res = lx; q.ldr = H; q.M = rows; q.N = H; q.K = H;
    q.w_scale = b0.o8.s; q.a_mx = lmx;
    mx = gemm_mx_supported(q, VSTAR_EPI_NONE);
    q.A = (const lp_t*)lact; q.lda = c.llm_mlp; q.W = (const lp_t*)b0.down8.W; q.K = c.llm_mlp; q.w_scale = b0.down8.s;
    mx = mx && gemm_mx_supported(q, VSTAR_EPI_NONE);
    q = GemmParams{};
    q.A = (const lp_t*)lq8; q.lda = H; q.W = (const lp_t*)b0.gate_up8.W; q.C = lact; q.ldc = c.llm_mlp; q.M = rows; q.N = b0.gate_up.N; q.K = H;
    q.a_scale = lsa; q.w_scale = b0.gate_up8.s; q.c_mx = lmx;
    mx = mx && gemm_mx_supported(q, VSTAR_EPI_SILU_MUL);
  }
  // Fully block-scaled chain (opt-in: VSTAR_W8A8_MX=2; the default, 1, stops at the step above): the residual stream ALSO leaves o_proj / down_proj
  // as a block-scaled fp8 copy with sum-of-squares partials (mx_none_epilogue), q|k|v and gate|up consume it with the RMSNorm folded
  // (weight into their fp8 W, 1 / rms as the per-row scale) — no rmsnorm_quant pass either.
  // Opt-in because it moves the search: on the config-5 search leg (bench.py, one 341-node tree) levels 0 / 1 / 2 follow the bf16 visit
  // order for 25 / 20 / 3 nodes and level 2 ends in another final box, at +1 % speed over level 1 (profiles/r06_w8a8_levels.txt).
  static const int mx_level = [] { const char* e = getenv("VSTAR_W8A8_MX"); return e ? atoi(e) : 1; }();
  bool chain = false;
  if (mx && mx_level >= 2 && H % 256 == 0) {
    GemmParams q{};
    LlmBlock& b0 = llm[0];
    q.A = (const lp_t*)latt; q.lda = H; q.W = (const lp_t*)b0.o8.W; q.C = lx; q.ldc = H; q.res = lx; q.ldr = H; q.M = rows; q.N = H; q.K = H;
    q.w_scale = b0.o8.s; q.a_mx = lmx; q.c_mx = lmxx; q.c8 = lq8; q.ldc8 = H; q.sumsq_out = lpart; q.sumsq_ld = H / 64;
    chain = gemm_mx_supported(q, VSTAR_EPI_NONE);
    q = GemmParams{};
    q.A = (const lp_t*)lq8; q.lda = H; q.W = (const lp_t*)b0.qkv8f.W; q.C = lqkv; q.ldc = 3 * H; q.M = rows; q.N = b0.qkv.N; q.K = H;
    q.a_scale = lr; q.w_scale = b0.qkv8f.s; q.a_mx = lmxx;
    if (fused_rope) { q.rope_cs = rope; q.rope_S = S; q.rope_cols = 2 * H; }
    chain = chain && gemm_mx_supported(q, VSTAR_EPI_NONE);
  }
  last_w8a8_mx = mx;
  last_w8a8_chain = chain ? 2 : mx ? 1 : 0;
  // Shared prefix (VSTAR_F_SHARE_PREFIX, psh_Lp > 0): the first Lp positions of every sequence are the same tokens, so they run
  // ONCE, as a sequence of their own in the rows [Lp, 2 Lp) of slot `nseq` of the activation buffers.  The linears see the
  // compact row set {rows [Lp, S) of every sequence} + {the prefix rows} through a periodic row map (group S - Lp, stride S,
  // offset Lp: the partial last group lands exactly on the prefix rows); row-wise kernels simply run over all buffer rows;
  // the attention runs unchanged over the full sequences after the prefix's q|k|v rows were copied to the head of each, plus
  // one small launch for the prefix sequence itself.
  const int Lp = w8 ? 0 : psh_Lp;
  const bool fold = fold_norms && !w8;
  const int Sr = S - Lp;
  const int M = Lp ? nseq * Sr + Lp : rows;             // rows of the linears
  const int nrows = Lp ? rows + 2 * Lp : rows;          // rows of the row-wise kernels (slot nseq: [0, 2 Lp))
  const size_t pre = (size_t)rows + Lp;                 // first buffer row of the prefix sequence
  struct MapGuard {                                      // the row map never outlives this call (early error returns included)
    vstar_engine* e;
    ~MapGuard() { e->map_group = 0; }
  } guard{this};
  auto set_map = [&](bool on) {
    map_group = (on && Lp) ? Sr : 0;
    map_gstride = S;
    map_off = Lp;
  };
  for (int i = 0; i < c.llm_layers; ++i) {
    LlmBlock& b = llm[i];
    set_map(true);
    if (chain) {
      // rows of lx as block-scaled fp8 (lq8 / lmxx) + their 1 / rms (lr): from the splice for the first block, from the previous
      // block's down_proj epilogue afterwards
      if (i == 0) {
        KCHK(quantize_rows_mx(lx, H, lq8, H, lmxx, rows, H, stream));
        KCHK(rms_rstd_rows(lx, rows, H, c.llm_rms_eps, lr, stream));
      } else {
        KCHK(rms_rstd_partials(lpart, H / 64, rows, H, c.llm_rms_eps, lr, stream));
      }
      RC(lin8(lq8, lr, b.qkv, b.qkv8f, lqkv, 3 * H, rows, VSTAR_EPI_NONE, nullptr, 0, fused_rope ? rope : nullptr, S, 2 * H, lmxx));
      if (!fused_rope) KCHK(attn_prepare(lqkv, rope, nseq, S, c.llm_heads, 128, stream, grp_R0, grp_Lc));
    } else if (w8) {
      KCHK(rmsnorm_quant_fp8(lx, b.in_norm, lq8, lsa, rows, H, c.llm_rms_eps, stream));
      RC(lin8(lq8, lsa, b.qkv, b.qkv8, lqkv, 3 * H, rows, VSTAR_EPI_NONE, nullptr, 0, fused_rope ? rope : nullptr, S, 2 * H));
      if (!fused_rope) KCHK(attn_prepare(lqkv, rope, nseq, S, c.llm_heads, 128, stream, grp_R0, grp_Lc));
    } else {
      // input RMSNorm: folded (statistics from x for the first block — the spliced embeddings — else from the partial sums the
      // previous block's down_proj epilogue wrote next to the residual stream) or the norm kernel
      GemmParams p{};
      if (fold) {
        if (i == 0) KCHK(rms_rstd_rows(lx, nrows, H, c.llm_rms_eps, lr, stream));
        else KCHK(rms_rstd_partials(lpart, H / 64, nrows, H, c.llm_rms_eps, lr, stream));
        p.A = lx; p.row_scale = lr;
      } else {
        KCHK(rmsnorm_lp(lx, b.in_norm, lh, nrows, H, c.llm_rms_eps, nullptr, stream));
        p.A = lh;
      }
      // q|k|v projection; RoPE rides in the GEMM epilogue when the 256^2 kernel takes the shape (else a separate pass)
      p.lda = H; p.W = b.qkv.W; p.C = lqkv; p.ldc = 3 * H; p.M = M; p.N = b.qkv.N; p.K = b.qkv.K;
      if (Lp) { p.a_group = p.c_group = Sr; p.a_gstride = p.c_gstride = S; p.a_off = p.c_off = Lp; }
      const bool fused = fused_rope && gemm256_eligible(p);
      if (fused) {
        p.rope_cs = rope; p.rope_S = Lp ? Sr : S; p.rope_cols = 2 * H; p.rope_R0 = grp_R0; p.rope_Lc = grp_Lc;
        if (Lp) { p.rope_pos0 = Lp; p.rope_tail = nseq * Sr; }
      }
      RC(gemm(p, VSTAR_EPI_NONE, false));
      if (!fused) {
        KCHK(attn_prepare(lqkv, rope, nseq, S, c.llm_heads, 128, stream, grp_R0, grp_Lc));     // (prefix rows of the sequences: overwritten below)
        if (Lp) KCHK(attn_prepare(lqkv + pre * 3 * H, rope, 1, Lp, c.llm_heads, 128, stream, 0, 0));
      }
    }
    if (Lp) {
      KCHK(bcast_rows(lqkv + pre * 3 * H, lqkv, nseq, S, Lp, 3 * H, 3 * H, stream));
      KCHK(attn_forward(lqkv + pre * 3 * H, latt + pre * H, 1, Lp, c.llm_heads, 128, 1, att_scale, stream, 0, 0));
    }
    if (mx && i + 1 < c.llm_layers) KCHK(attn_forward_mx(lqkv, (uint8_t*)latt, lmx, nseq, S, c.llm_heads, att_scale, stream, grp_R0, grp_Lc));
    else KCHK(attn_forward(lqkv, latt, nseq, S, c.llm_heads, 128, 1, att_scale, stream, grp_R0, grp_Lc));
    if (i + 1 == c.llm_layers) {
      // Last block: only the [LOC]-1 row and the verify rows are ever read (VSM.py:465-473), and every op after the
      // attention is row-wise, so o_proj / MLP run on those gathered rows only (row-wise ops: bit-identical).
      // (W8A8 mode: these few rows stay on the bf16 weights — the weight-bound regime gains nothing from fp8 MFMA.)
      set_map(false);
      KCHK(gather_rows(latt, d_rowidx, sel_att, nsel, H, stream));
      KCHK(gather_rows(lx, d_rowidx, sel_x, nsel, H, stream));
      RC(lin(sel_att, H, b.o, sel_x, H, nsel, VSTAR_EPI_NONE, sel_x, H));
      KCHK(rmsnorm_lp(sel_x, b.post_norm, sel_h, nsel, H, c.llm_rms_eps, nullptr, stream));
      RC(lin(sel_h, H, b.gate_up, sel_act, c.llm_mlp, nsel, VSTAR_EPI_SILU_MUL));
      RC(lin(sel_act, c.llm_mlp, b.down, sel_x, H, nsel, VSTAR_EPI_NONE, sel_x, H));
      break;
    }
    if (chain) {
      RC(lin8((const uint8_t*)latt, nullptr, b.o, b.o8, lx, H, rows, VSTAR_EPI_NONE, lx, H, nullptr, 0, 0, lmx, lmxx, lq8, lpart));
      KCHK(rms_rstd_partials(lpart, H / 64, rows, H, c.llm_rms_eps, lr, stream));
      RC(lin8(lq8, lr, b.gate_up, b.gate_up8f, lact, c.llm_mlp, rows, VSTAR_EPI_SILU_MUL, nullptr, 0, nullptr, 0, 0, lmxx, lmx));
      RC(lin8((const uint8_t*)lact, nullptr, b.down, b.down8, lx, H, rows, VSTAR_EPI_NONE, lx, H, nullptr, 0, 0, lmx, lmxx, lq8, lpart));
    } else if (mx) {
      RC(lin8((const uint8_t*)latt, nullptr, b.o, b.o8, lx, H, rows, VSTAR_EPI_NONE, lx, H, nullptr, 0, 0, lmx, nullptr));
      KCHK(rmsnorm_quant_fp8(lx, b.post_norm, lq8, lsa, rows, H, c.llm_rms_eps, stream));
      RC(lin8(lq8, lsa, b.gate_up, b.gate_up8, lact, c.llm_mlp, rows, VSTAR_EPI_SILU_MUL, nullptr, 0, nullptr, 0, 0, nullptr, lmx));
      RC(lin8((const uint8_t*)lact, nullptr, b.down, b.down8, lx, H, rows, VSTAR_EPI_NONE, lx, H, nullptr, 0, 0, lmx, nullptr));
    } else if (w8) {
      KCHK(quantize_rows_fp8(latt, H, lq8, H, lsa, rows, H, stream));
      RC(lin8(lq8, lsa, b.o, b.o8, lx, H, rows, VSTAR_EPI_NONE, lx, H));
      KCHK(rmsnorm_quant_fp8(lx, b.post_norm, lq8, lsa, rows, H, c.llm_rms_eps, stream));
      RC(lin8(lq8, lsa, b.gate_up, b.gate_up8, lact, c.llm_mlp, rows, VSTAR_EPI_SILU_MUL, nullptr, 0));
      KCHK(quantize_rows_fp8(lact, c.llm_mlp, lq8, c.llm_mlp, lsa, rows, c.llm_mlp, stream));
      RC(lin8(lq8, lsa, b.down, b.down8, lx, H, rows, VSTAR_EPI_NONE, lx, H));
    } else {
      if (fold) {
        next_sumsq = lpart; next_sumsq_ld = H / 64;
        RC(lin(latt, H, b.o, lx, H, M, VSTAR_EPI_NONE, lx, H));
        KCHK(rms_rstd_partials(lpart, H / 64, nrows, H, c.llm_rms_eps, lr, stream));
        next_row_scale = lr;
        RC(lin(lx, H, b.gate_up, lact, c.llm_mlp, M, VSTAR_EPI_SILU_MUL));
        next_sumsq = lpart; next_sumsq_ld = H / 64;
        RC(lin(lact, c.llm_mlp, b.down, lx, H, M, VSTAR_EPI_NONE, lx, H));
      } else {
        RC(lin(latt, H, b.o, lx, H, M, VSTAR_EPI_NONE, lx, H));
        KCHK(rmsnorm_lp(lx, b.post_norm, lh, nrows, H, c.llm_rms_eps, nullptr, stream));
        RC(lin(lh, H, b.gate_up, lact, c.llm_mlp, M, VSTAR_EPI_SILU_MUL));
        RC(lin(lact, c.llm_mlp, b.down, lx, H, M, VSTAR_EPI_NONE, lx, H));
      }
    }
  }
  return 0;
}

// ---- a6/a7/a8: final norm on the needed rows, lm_head argmax at the verify rows, text_hidden_fcs on the [LOC]-1 rows
// (VSM.py:120-140,465-486).  sel_x rows: [0, nrec) the [LOC]-1 states, then nrec * n_verify verify rows.
int vstar_engine::llm_heads(int nrec, int n_verify) {
  const vstar_config& c = cfg;
  const int H = c.llm_hidden;
  const int nsel = nrec * (1 + n_verify);
  KCHK(rmsnorm_lp(sel_x, final_norm, hsel, nsel, H, c.llm_rms_eps, nullptr, stream));
  if (n_verify > 0) {
    RC(lin(hsel + (size_t)nrec * H, H, lm_head, vlogits, c.llm_vocab, nrec * n_verify, VSTAR_EPI_NONE, nullptr, 0, true));
    KCHK(argmax_rows(vlogits, nrec * n_verify, c.llm_vocab, c.llm_vocab, d_argmax, 1, stream));
  }
  RC(lin(hsel, H, det0, fc_tmp, H, nrec, VSTAR_EPI_RELU));
  RC(lin(fc_tmp, H, det1, emb_det, det1.N, nrec));
  RC(lin(hsel, H, seg0, fc_tmp, H, nrec, VSTAR_EPI_RELU));
  RC(lin(fc_tmp, H, seg1, emb_seg, 256, nrec));
  return 0;
}

// ---- a9-a11: OWL-ViT tower on Bimg crops, class/box heads and the SAM-style mask head for nrec records; record n reads the image
// features of crop n / img_div (grouped scoring: img_div prompts per crop; 1 otherwise)
int vstar_engine::owl_heads_sam(const lp_t* opix, int Bimg, int nrec, int img_div) {
  RC(owl_features(opix, Bimg));
  return owl_finish(Bimg, nrec, img_div);
}

// Armed after fork_owl has queued work on stream2: whichever way the scoring call leaves (an RC() failure of the main path included),
// stream2 is drained first, so the next call's stage_pixels / preprocessing cannot overwrite d_owl_pix or the OWL scratch while the
// side stream still reads them.  The success path disarms it after its own event wait.
struct OwlJoinGuard {
  hipStream_t s2;
  bool armed;
  ~OwlJoinGuard() {
    if (!armed || !s2) return;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s2, &st) == hipSuccess && st != hipStreamCaptureStatusNone) return;   // inside a graph capture: nothing runs yet
    (void)hipStreamSynchronize(s2);
  }
};

bool vstar_engine::fork_owl(const lp_t* opix, int Bimg, int* rc) {
  *rc = 0;
  const bool want = owl_overlap >= 0 ? owl_overlap != 0 : Bimg <= 8;
  if (!want || profile || !stream2) return false;
  // everything already queued on the main stream (pixel staging, the previous call's copy-out of d_results) precedes the fork
  if (hipEventRecord(ev_fork, stream) != hipSuccess || hipStreamWaitEvent(stream2, ev_fork, 0) != hipSuccess) {
    set_error("owl overlap: fork failed"); *rc = VSTAR_ERR_HIP; return false;
  }
  std::swap(stream, stream2);                  // the launch helpers enqueue on `stream`
  *rc = owl_features(opix, Bimg);
  std::swap(stream, stream2);
  if (*rc == 0 && hipEventRecord(ev_join, stream2) != hipSuccess) { set_error("owl overlap: join event failed"); *rc = VSTAR_ERR_HIP; }
  return true;
}

int vstar_engine::owl_features(const lp_t* opix, int Bimg) {
  const vstar_config& c = cfg;
  const int OH = c.owl_hidden, NP = owl.P, irow = Bimg * NP;
  // ---- a9: OWL-ViT tower + get_visual_embs (owlvit.py:121-148) ----
  RC(run_tower(owl, opix, Bimg));
  KCHK(layernorm_lp(owl.x, owl_post_g, owl_post_b, owl.h, Bimg * owl.N, OH, 1e-5f, nullptr, 0, stream));
  KCHK(owl_cls_mul(owl.h, owl_feats_pre, Bimg, owl.N, OH, stream));
  KCHK(layernorm_lp(owl_feats_pre, owl_ln_g, owl_ln_b, owl_feats, irow, OH, 1e-5f, nullptr, 0, stream));
  // ---- a10: class + box heads (owlvit.py:150-170) ----
  const int cld = cls_fused.N + 2;
  RC(lin(owl_feats, OH, cls_fused, cls_emb, cld, irow, VSTAR_EPI_NONE, nullptr, 0, true));
  RC(lin(owl_feats, OH, box0, box_t0, OH, irow, VSTAR_EPI_GELU));
  RC(lin(box_t0, OH, box1, box_t1, OH, irow, VSTAR_EPI_GELU));
  RC(lin(box_t1, OH, box2, box_raw, 4, irow, VSTAR_EPI_NONE, nullptr, 0, true));
  // ---- a11 (first step): visual_projection (VSM.py:515-517) ----
  RC(lin(owl_feats, OH, vis_proj, s_src, 256, irow));
  return 0;
}

int vstar_engine::owl_finish(int Bimg, int nrec, int img_div) {
  const vstar_config& c = cfg;
  const int B = nrec;
  const int NP = owl.P, prow = nrec * NP;
  const int rstride = (int)(sizeof(vstar_result) / 4);
  float* res_f = (float*)d_results;
  const int cld = cls_fused.N + 2;
  (void)Bimg;
  // ---- a10: class logits against the det embedding, box finisher (owlvit.py:150-170) ----
  KCHK(owl_class_logits(cls_emb, cld, c.owl_query_dim, emb_det, res_f + offsetof(vstar_result, pred_logits) / 4, rstride, nrec,
                        NP, stream, img_div));
  KCHK(owl_box_finish(box_raw, 4, res_f + offsetof(vstar_result, pred_boxes) / 4, rstride, nrec, owl.grid, stream, img_div));
  // ---- a11: prompt encoder + two-way transformer + upscaling (VSM.py:518-533) ----
  if (img_div <= 1) {
    KCHK(add_bcast(s_src, no_mask, s_keys, prow, 256, 1, stream));              // src = image_embeddings + dense (no_mask_embed)
  } else {
    KCHK(add_bcast_repeat(s_src, no_mask, s_keys, nrec, img_div, NP, 256, stream));   // one private copy of the keys per prompt
  }
  hipLaunchKernelGGL(sam_tokens_kernel, dim3((B * 6 * 256 + 255) / 256), dim3(256), 0, stream, iou_token, mask_tokens,
                     emb_seg, s_tok0, B);
  KCHK(hipGetLastError());
  const int T = 6, trow = B * T;
  HIPCHK(hipMemcpyAsync(s_q, s_tok0, (size_t)trow * 256 * 2, hipMemcpyDeviceToDevice, stream));
  for (int i = 0; i < 2; ++i) {
    SamLayer& Ly = sam_layers[i];
    // (1) token self attention
    if (i == 0) {
      RC(sam_attn(Ly.self_attn, s_q, T, s_q, s_q, T, B, s_ta, s_tb, s_tc, s_td, s_qpe, nullptr));
    } else {
      KCHK(add_bcast(s_q, s_tok0, s_mlp, trow, 256, trow, stream));           // q = queries + query_pe
      RC(sam_attn(Ly.self_attn, s_mlp, T, s_mlp, s_q, T, B, s_ta, s_tb, s_tc, s_td, s_qpe, s_q));
    }
    KCHK(layernorm_lp(s_qpe, Ly.n1g, Ly.n1b, s_q, trow, 256, 1e-5f, nullptr, 0, stream));
    // (2) tokens -> image cross attention
    KCHK(add_bcast(s_q, s_tok0, s_qpe, trow, 256, trow, stream));             // q = queries + query_pe
    KCHK(add_bcast(s_keys, dense_pe, s_kpe, prow, 256, NP, stream));          // k = keys + key_pe
    RC(sam_attn(Ly.t2i, s_qpe, T, s_kpe, s_keys, NP, B, s_ta, s_ia, s_ib, s_td, s_tb, s_q));
    KCHK(layernorm_lp(s_tb, Ly.n2g, Ly.n2b, s_q, trow, 256, 1e-5f, nullptr, 0, stream));
    // (3) MLP on tokens
    RC(lin(s_q, 256, Ly.lin1, s_mlp, Ly.lin1.N, trow, VSTAR_EPI_RELU));
    RC(lin(s_mlp, Ly.lin1.N, Ly.lin2, s_tb, 256, trow, VSTAR_EPI_NONE, s_q, 256));
    KCHK(layernorm_lp(s_tb, Ly.n3g, Ly.n3b, s_q, trow, 256, 1e-5f, nullptr, 0, stream));
    // (4) image -> tokens cross attention (q = keys + key_pe, k = queries + query_pe, v = queries)
    KCHK(add_bcast(s_q, s_tok0, s_qpe, trow, 256, trow, stream));
    RC(sam_attn(Ly.i2t, s_kpe, NP, s_qpe, s_q, T, B, s_ia, s_ta, s_tb, s_ib, s_ic, s_keys));
    KCHK(layernorm_lp(s_ic, Ly.n4g, Ly.n4b, s_keys, prow, 256, 1e-5f, nullptr, 0, stream));
  }
  KCHK(add_bcast(s_q, s_tok0, s_qpe, trow, 256, trow, stream));
  KCHK(add_bcast(s_keys, dense_pe, s_kpe, prow, 256, NP, stream));
  RC(sam_attn(sam_final, s_qpe, T, s_kpe, s_keys, NP, B, s_ta, s_ia, s_ib, s_td, s_tb, s_q));
  KCHK(layernorm_lp(s_tb, sam_nfg, sam_nfb, s_q, trow, 256, 1e-5f, nullptr, 0, stream));
  // hypernetwork MLP 0 on mask token 0 (= token row 1)
  KCHK(gather_rows(s_q, d_tokidx, s_hyp_in, B, 256, stream));
  RC(lin(s_hyp_in, 256, hyp0, s_hyp_a, 256, B, VSTAR_EPI_RELU));
  RC(lin(s_hyp_a, 256, hyp1, s_hyp_b, 256, B, VSTAR_EPI_RELU));
  RC(lin(s_hyp_b, 256, hyp2, s_hyper, 32, B));
  // output_upscaling: Upsample(256->64) LN2d GELU Upsample(64->32) GELU (mask_decoder.py:78-84)
  KCHK(upsample2x_im2col3x3(s_keys, s_col1, B, 48, 48, 256, stream));
  RC(lin(s_col1, 2304, conv1, s_c1, 64, B * 96 * 96));
  KCHK(layernorm_lp(s_c1, ln2d_g, ln2d_b, s_c1n, B * 96 * 96, 64, 1e-6f, nullptr, 1, stream));
  KCHK(upsample2x_im2col3x3(s_c1n, s_col2, B, 96, 96, 64, stream));
  RC(lin(s_col2, 576, conv2, s_c2, 32, B * 192 * 192, VSTAR_EPI_GELU));
  KCHK(hyper_mask(s_hyper, s_c2, res_f + offsetof(vstar_result, lowres_mask) / 4, rstride, B, 192 * 192, 32, stream));
  return 0;
}

int vstar_engine::finish_records(int nrec, int n_verify, unsigned flags, vstar_result* out) {
  if (n_verify > 0) {   // verify argmax -> records
    HIPCHK(hipMemcpy2DAsync((char*)d_results + offsetof(vstar_result, tf_argmax), sizeof(vstar_result), d_argmax,
                            (size_t)n_verify * 4, (size_t)n_verify * 4, nrec, hipMemcpyDeviceToDevice, stream));
  }
  if (out) {
    HIPCHK(hipMemcpyAsync(out, d_results, sizeof(vstar_result) * nrec,
                          (flags & VSTAR_F_DEVICE_OUTPUT) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, stream));
  }
  if (!(flags & VSTAR_F_NO_SYNC)) {
    HIPCHK(hipStreamSynchronize(stream));
    collect_profile();
  }
  return 0;
}

int vstar_engine::score(int B, const lp_t* clip_pix, const lp_t* owl_pix, const int32_t* ids, int L,
                        const int32_t* loc_pos, const int32_t* verify_pos, int n_verify, unsigned flags,
                        vstar_result* out) {
  if (!finalized) { set_error("vstar_finalize_weights has not been called"); return VSTAR_ERR_STATE; }
  const vstar_config& c = cfg;
  if (B <= 0 || B > c.max_batch) { set_error("B out of range"); return VSTAR_ERR_INVALID; }
  if (L < 2 || L > c.max_text_len) { set_error("L out of range"); return VSTAR_ERR_INVALID; }
  if (n_verify < 0 || n_verify > VSTAR_MAX_VERIFY) { set_error("n_verify out of range"); return VSTAR_ERR_INVALID; }
  const bool internal_pix = flags & VSTAR_F_INTERNAL_PIXELS;
  if (!ids || !loc_pos || (!clip_pix && !internal_pix) || (n_verify && !verify_pos)) { set_error("null input"); return VSTAR_ERR_INVALID; }
  const bool skip_owl = flags & VSTAR_F_SKIP_OWL;
  if (!skip_owl && !owl_pix && !internal_pix) { set_error("null owl_pix"); return VSTAR_ERR_INVALID; }
  HIPCHK(hipSetDevice(device));
  // exactly one IMAGE_TOKEN_INDEX (-200) per row, same column in every row
  int img_col = -1;
  for (int b = 0; b < B; ++b) {
    int cnt = 0, col = -1;
    for (int j = 0; j < L; ++j)
      if (ids[(size_t)b * L + j] == -200) { cnt++; col = j; }
    if (cnt != 1) { set_error("each ids row must contain exactly one -200 image token"); return VSTAR_ERR_INVALID; }
    if (b == 0) img_col = col;
    else if (col != img_col) { set_error("image token column differs across the batch"); return VSTAR_ERR_INVALID; }
  }
  const int P = clip.P, S = L - 1 + P, H = c.llm_hidden;
  std::vector<int32_t>& rowidx = h_rowidx;   // member: must outlive the async copy
  rowidx.assign((size_t)B * (1 + n_verify), 0);
  for (int b = 0; b < B; ++b) {
    if (loc_pos[b] < 0 || loc_pos[b] >= S) { set_error("loc_pos out of range"); return VSTAR_ERR_INVALID; }
    rowidx[b] = b * S + loc_pos[b];
    for (int v = 0; v < n_verify; ++v) {
      const int pv = verify_pos[(size_t)b * n_verify + v];
      if (pv < 0 || pv >= S) { set_error("verify_pos out of range"); return VSTAR_ERR_INVALID; }
      rowidx[(size_t)B + (size_t)b * n_verify + v] = b * S + pv;
    }
  }
  const lp_t *cpix = nullptr, *opix = nullptr;
  RC(stage_pixels(B, clip_pix, owl_pix, flags, skip_owl, &cpix, &opix));
  // VSTAR_F_SHARE_PREFIX: the text before <image> (>= 16 tokens, identical in every row, and short enough for the prefix
  // sequence to sit in [Lp, 2 Lp) of one sequence slot) is computed once; its embeddings are those of sequence 0
  int psh = 0;
  if ((flags & VSTAR_F_SHARE_PREFIX) && !c.llm_w8a8 && img_col >= 16 && 2 * img_col <= S) {
    bool same = true;
    for (int b = 1; b < B && same; ++b) same = memcmp(ids, ids + (size_t)b * L, (size_t)img_col * 4) == 0;
    if (same) psh = img_col;
  }
  last_B = B; last_S = S;
  const bool graphs_on = score_graph_on;
  const bool dev_pix = (flags & (VSTAR_F_INTERNAL_PIXELS | VSTAR_F_DEVICE_INPUTS)) != 0;
  grp_R0 = grp_Lc = 0;      // host-side state of the pass: reset HERE, not in score_body — a replayed graph never runs that code
  // Not for VSTAR_F_NO_SYNC callers: the graph's H2D nodes read the single pinned staging buffers asynchronously, so a second
  // un-synchronised call would overwrite them before the first graph's copy has run (the eager path stages pageable `ids` at call
  // time and has no such window).
  if (graphs_on && dev_pix && B <= score_graph_max_B && !profile && h_ids_pin && h_rowidx_pin && !(flags & VSTAR_F_NO_SYNC)) {
    // per-call data through the pinned staging buffers (this call synchronises before it returns, see the condition above)
    memcpy(h_ids_pin, ids, (size_t)B * L * 4);
    memcpy(h_rowidx_pin, rowidx.data(), rowidx.size() * 4);
    const ScoreSig sig{B, L, n_verify, img_col, psh, flags & ~(unsigned)VSTAR_F_NO_SYNC, cpix, opix};
    auto it = score_graphs.find(sig);
    if (it != score_graphs.end() && it->second) {
      HIPCHK(hipGraphLaunch(it->second, stream));
      ++score_graph_replays;
      return finish_records(B, n_verify, flags, out);
    }
    if (it != score_graphs.end()) {                       // second call with this signature: capture it
      hipGraph_t graph = nullptr;
      if (hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
        const int rc = score_body(B, L, n_verify, flags, img_col, psh, cpix, opix, skip_owl, h_ids_pin, h_rowidx_pin, rowidx.size());
        const hipError_t ce = hipStreamEndCapture(stream, &graph);
        hipGraphExec_t exec = nullptr;
        if (rc == 0 && ce == hipSuccess && graph && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess && exec) {
          hipGraphDestroy(graph);
          it->second = exec;
          HIPCHK(hipGraphLaunch(exec, stream));
          ++score_graph_replays;
          return finish_records(B, n_verify, flags, out);
        }
        if (graph) hipGraphDestroy(graph);
        (void)hipGetLastError();
        score_graphs.erase(it);                           // capture refused: this signature stays on the eager path
        score_graph_max_B = 0;                            // ... and so does everything else (one failure is a property of the runtime)
        set_error("");
      } else {
        (void)hipGetLastError();
        score_graph_max_B = 0;
      }
    } else {
      score_graphs[sig] = nullptr;                        // first call: eager (warm-up), remember the signature
    }
    RC(score_body(B, L, n_verify, flags, img_col, psh, cpix, opix, skip_owl, h_ids_pin, h_rowidx_pin, rowidx.size()));
    return finish_records(B, n_verify, flags, out);
  }
  RC(score_body(B, L, n_verify, flags, img_col, psh, cpix, opix, skip_owl, ids, rowidx.data(), rowidx.size()));
  return finish_records(B, n_verify, flags, out);
}

// Everything a scoring step ENQUEUES (no host synchronisation, no host decision that the arguments do not already carry): what
// score() runs eagerly and what it captures into a hipGraph for small batches.
int vstar_engine::score_body(int B, int L, int n_verify, unsigned flags, int img_col, int psh, const lp_t* cpix, const lp_t* opix,
                             bool skip_owl, const int32_t* ids_src, const int32_t* rowidx_src, size_t n_rowidx) {
  const vstar_config& c = cfg;
  const int P = clip.P, S = L - 1 + P, H = c.llm_hidden;
  HIPCHK(hipMemcpyAsync(d_ids, ids_src, (size_t)B * L * 4, hipMemcpyHostToDevice, stream));
  HIPCHK(hipMemcpyAsync(d_rowidx, rowidx_src, n_rowidx * 4, hipMemcpyHostToDevice, stream));
  int frc_owl = 0;
  const bool forked = !skip_owl && fork_owl(opix, B, &frc_owl);       // OWL-ViT side of the graph on stream2 (small batches)
  OwlJoinGuard join_guard{stream2, forked || frc_owl != 0};
  RC(frc_owl);
  grp_R0 = grp_Lc = 0;

  // ---- a2: CLIP tower (clip_encoder.py:31-60) ----
  RC(run_tower(clip, cpix, B));
  // ---- a3+a4: mm_projector on the patch rows, written straight into the spliced LLaMA input (llava_arch.py:93-96,185-208)
  {
    GemmParams p{};
    p.A = clip.x; p.lda = clip.hidden; p.a_group = P; p.a_gstride = clip.N; p.a_off = 1;
    p.W = projector.W; p.bias = projector.b; p.res = nullptr;
    p.C = lx; p.ldc = H; p.c_group = P; p.c_gstride = S; p.c_off = img_col;
    p.M = B * P; p.N = projector.N; p.K = projector.K;
    RC(gemm(p, VSTAR_EPI_NONE, false));
  }
  KCHK(llm_embed_text(d_ids, L, img_col, P, embed, c.llm_vocab, lx, B, H, stream));
  psh_Lp = psh;
  if (psh_Lp)
    HIPCHK(hipMemcpyAsync(lx + ((size_t)B * S + psh_Lp) * H, lx, (size_t)psh_Lp * H * sizeof(lp_t), hipMemcpyDeviceToDevice, stream));
  const int frc = llm_forward(B, S, B * (1 + n_verify));
  psh_Lp = 0;
  RC(frc);
  RC(llm_heads(B, n_verify));
  if (!skip_owl) {
    if (forked) {
      HIPCHK(hipStreamWaitEvent(stream, ev_join, 0));
      join_guard.armed = false;
      RC(owl_finish(B, B, 1));
    } else {
      RC(owl_heads_sam(opix, B, B, 1));
    }
  }
  return 0;
}

// Grouped scoring: G crops, each scored for T prompts whose first Lc spliced positions coincide (system prompt, image tokens and
// the common start of the question).  The shared positions run through LLaMA ONCE per crop; each prompt adds one 32-row suffix
// block that attends to them (under the causal mask the shared positions' states do not depend on what follows, so every
// record equals the one vstar_vsm_score_batch computes for that (crop, prompt) pair up to the summation order of the attention).
// Sequence layout per crop: rows [0, Lc) shared | [Lc, R0) zero padding, R0 = round_up(Lc, 128) | T blocks of 32 rows.
int vstar_engine::score_grouped(int G, int T, const lp_t* clip_pix, const lp_t* owl_pix, const int32_t* prefix_ids, int Lp,
                                const int32_t* suffix_ids, int Ls, const int32_t* loc_in_suffix, const int32_t* verify_in_suffix,
                                int n_verify, unsigned flags, vstar_result* out) {
  if (!finalized) { set_error("vstar_finalize_weights has not been called"); return VSTAR_ERR_STATE; }
  const vstar_config& c = cfg;
  const int nrec = G * T;
  if (G <= 0 || T <= 0 || nrec > c.max_batch) { set_error("G * T out of range (max_batch records per call)"); return VSTAR_ERR_INVALID; }
  if (Ls < 1 || Ls > 32 || Lp < 2) { set_error("suffix length must be 1..32 tokens"); return VSTAR_ERR_INVALID; }
  if (n_verify < 0 || n_verify > VSTAR_MAX_VERIFY) { set_error("n_verify out of range"); return VSTAR_ERR_INVALID; }
  const bool internal_pix = flags & VSTAR_F_INTERNAL_PIXELS;
  if (!prefix_ids || !suffix_ids || !loc_in_suffix || (!clip_pix && !internal_pix) || (!owl_pix && !internal_pix) ||
      (n_verify && !verify_in_suffix)) { set_error("null input"); return VSTAR_ERR_INVALID; }
  if (flags & VSTAR_F_SKIP_OWL) { set_error("grouped scoring always runs the heads"); return VSTAR_ERR_INVALID; }
  HIPCHK(hipSetDevice(device));
  int img_col = -1;
  for (int j = 0; j < Lp; ++j)
    if (prefix_ids[j] == -200) { if (img_col >= 0) { set_error("more than one -200 in the shared prefix"); return VSTAR_ERR_INVALID; } img_col = j; }
  if (img_col < 0) { set_error("the shared prefix must contain the -200 image token"); return VSTAR_ERR_INVALID; }
  const int P = clip.P, H = c.llm_hidden;
  const int Lc = Lp - 1 + P, R0 = (Lc + 127) / 128 * 128, S = R0 + 32 * T;
  if ((size_t)G * S > (size_t)c.max_batch * Smax || R0 > Smax + 128) {
    set_error("grouped batch exceeds the LLaMA workspace (G * (round_up(Lc,128) + 32 T) rows)");
    return VSTAR_ERR_INVALID;
  }
  // ---- embed_rows sources + the rows whose states are needed ----
  h_src.assign((size_t)G * S, INT32_MIN);
  std::vector<int32_t>& rowidx = h_rowidx;
  rowidx.assign((size_t)nrec * (1 + n_verify), 0);
  for (int g = 0; g < G; ++g) {
    int32_t* row = h_src.data() + (size_t)g * S;
    int r = 0;
    for (int j = 0; j < Lp; ++j) {
      if (j == img_col) { for (int q = 0; q < P; ++q) row[r++] = -(1 + g * P + q); }
      else {
        const int id = prefix_ids[j];
        if (id < 0 || id >= c.llm_vocab) { set_error("token id out of range in the shared prefix"); return VSTAR_ERR_INVALID; }
        row[r++] = id;
      }
    }
    for (int t = 0; t < T; ++t) {
      const int n = g * T + t;
      for (int j = 0; j < Ls; ++j) {
        const int id = suffix_ids[((size_t)n) * Ls + j];
        if (id < 0 || id >= c.llm_vocab) { set_error("token id out of range in a suffix"); return VSTAR_ERR_INVALID; }
        row[R0 + 32 * t + j] = id;
      }
      const int lp = loc_in_suffix[n];
      if (lp < 0 || lp >= Ls) { set_error("loc_in_suffix out of range"); return VSTAR_ERR_INVALID; }
      rowidx[n] = g * S + R0 + 32 * t + lp;
      for (int v = 0; v < n_verify; ++v) {
        const int pv = verify_in_suffix[(size_t)n * n_verify + v];
        if (pv < 0 || pv >= Ls) { set_error("verify_in_suffix out of range"); return VSTAR_ERR_INVALID; }
        rowidx[(size_t)nrec + (size_t)n * n_verify + v] = g * S + R0 + 32 * t + pv;
      }
    }
  }
  HIPCHK(hipMemcpyAsync(d_src, h_src.data(), h_src.size() * 4, hipMemcpyHostToDevice, stream));
  HIPCHK(hipMemcpyAsync(d_rowidx, rowidx.data(), rowidx.size() * 4, hipMemcpyHostToDevice, stream));
  const lp_t *cpix = nullptr, *opix = nullptr;
  RC(stage_pixels(G, clip_pix, owl_pix, flags, false, &cpix, &opix));
  int frc_owl = 0;
  const bool forked = fork_owl(opix, G, &frc_owl);
  OwlJoinGuard join_guard{stream2, forked || frc_owl != 0};
  RC(frc_owl);
  last_B = nrec; last_S = S;
  // ---- a2 + a3: CLIP tower and projector for the G crops -> feature table; a4: splice by row sources ----
  RC(run_tower(clip, cpix, G));
  {
    GemmParams p{};
    p.A = clip.x; p.lda = clip.hidden; p.a_group = P; p.a_gstride = clip.N; p.a_off = 1;
    p.W = projector.W; p.bias = projector.b; p.C = grp_feats; p.ldc = H; p.M = G * P; p.N = projector.N; p.K = projector.K;
    RC(gemm(p, VSTAR_EPI_NONE, false));
  }
  KCHK(embed_rows(d_src, embed, c.llm_vocab, grp_feats, (int64_t)G * P, lx, G * S, H, stream));
  grp_R0 = R0; grp_Lc = Lc;
  const int rc = llm_forward(G, S, nrec * (1 + n_verify));
  grp_R0 = grp_Lc = 0;
  RC(rc);
  RC(llm_heads(nrec, n_verify));
  if (forked) {
    HIPCHK(hipStreamWaitEvent(stream, ev_join, 0));
    join_guard.armed = false;
    RC(owl_finish(G, nrec, T));
  } else {
    RC(owl_heads_sam(opix, G, nrec, T));
  }
  return finish_records(nrec, n_verify, flags, out);
}

// VSM.inference(mode='vqa'): greedy decode of one crop, KV-cached (see include/vstar_hip.h)
int vstar_engine::generate(const lp_t* clip_pix, const int32_t* ids, int L, int max_new, int eos_id, unsigned flags,
                           int32_t* out_ids, int32_t* n_out) {
  if (!finalized) { set_error("weights not finalized"); return VSTAR_ERR_STATE; }
  const vstar_config& c = cfg;
  if (!clip_pix || !ids || !out_ids || !n_out || L < 2 || max_new < 1) { set_error("vstar_vsm_generate: bad argument"); return VSTAR_ERR_INVALID; }
  HIPCHK(hipSetDevice(device));
  const int H = c.llm_hidden, P = clip.P;
  const int ctx = (P + c.max_text_len + 255) / 64 * 64;
  if (!gen.ready) {
    LlmCachedCfg rc;
    rc.hidden = H; rc.heads = c.llm_heads; rc.mlp = c.llm_mlp; rc.layers = c.llm_layers; rc.vocab = c.llm_vocab;
    rc.rms_eps = c.llm_rms_eps; rc.rope_theta = c.llm_rope_theta;
    rc.max_slots = 1; rc.max_ctx = ctx; rc.max_rows = ctx;
    RC(dalloc(&gen_feats, (size_t)P * H));
    RC(gen.init(this, rc, embed, &llm, final_norm, &lm_head));
    gen.feats = gen_feats;
    gen.n_feat_rows = P;
  }
  int img = -1;
  for (int i = 0; i < L; ++i)
    if (ids[i] == -200) { if (img >= 0) { set_error("more than one -200 in input_ids"); return VSTAR_ERR_INVALID; } img = i; }
  if (img < 0) { set_error("input_ids hold no image token (-200)"); return VSTAR_ERR_INVALID; }
  if (L - 1 + P + max_new > ctx) { set_error("prompt + max_new_tokens exceed the decode context"); return VSTAR_ERR_INVALID; }
  // ---- CLIP tower + mm_projector -> feature rows ----
  const lp_t* cpix = clip_pix;
  if (!(flags & VSTAR_F_DEVICE_INPUTS)) {
    const size_t cn = (size_t)3 * c.clip_image_size * c.clip_image_size;
    HIPCHK(hipMemcpyAsync(d_clip_pix, clip_pix, cn * 2, hipMemcpyHostToDevice, stream));
    cpix = d_clip_pix;
  }
  RC(run_tower(clip, cpix, 1));
  {
    GemmParams p{};
    p.A = clip.x; p.lda = c.clip_hidden; p.a_group = P; p.a_gstride = clip.N; p.a_off = 1;
    p.W = projector.W; p.bias = projector.b; p.C = gen_feats; p.ldc = H; p.M = P; p.N = projector.N; p.K = projector.K;
    RC(gemm(p, VSTAR_EPI_NONE, false));
  }
  // ---- prefill the spliced prompt, then one decode step per new token ----
  std::vector<int32_t> rows;
  rows.reserve((size_t)L - 1 + P);
  for (int i = 0; i < L; ++i) {
    if (i == img) for (int j = 0; j < P; ++j) rows.push_back(-(1 + j));
    else rows.push_back(ids[i]);
  }
  int32_t row_off[2] = {0, (int32_t)rows.size()}, slot = 0, past = 0, want = (int32_t)rows.size() - 1, nxt = 0;
  RC(gen.forward(1, row_off, rows.data(), &slot, &slot, &past, 1, &want, nullptr, &nxt));
  int n = 0;
  past = (int32_t)rows.size();
  {
    // the decode loop as a replayed hipGraph (llm_cached.hpp): same kernels, no per-token host round trip
    bool used = false;
    RC(gen.decode_greedy_graph(nxt, past, slot, max_new, eos_id, out_ids, &n, &used));
    if (used) { *n_out = n; return 0; }
    n = 0;
  }
  for (;;) {
    out_ids[n++] = nxt;
    if (nxt == eos_id || n >= max_new) break;
    int32_t one_off[2] = {0, 1}, w0 = 0, tok = nxt;
    RC(gen.forward(1, one_off, &tok, &slot, &slot, &past, 1, &w0, nullptr, &nxt));
    ++past;
  }
  *n_out = n;
  return 0;
}


// =============================================== C ABI ===============================================
extern "C" {

int vstar_create(const vstar_config* cfg, int device, vstar_handle** out) {
  if (!cfg || !out) { tls_error() = "null argument"; return VSTAR_ERR_INVALID; }
  if (cfg->abi_version != VSTAR_ABI_VERSION) { tls_error() = "ABI version mismatch"; return VSTAR_ERR_INVALID; }
  if (cfg->max_batch <= 0 || cfg->max_text_len < 2) { tls_error() = "bad limits"; return VSTAR_ERR_INVALID; }
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || device < 0 || device >= n) {
    tls_error() = "no such HIP device (libvstar_hip has no CPU fallback)";
    return VSTAR_ERR_HIP;
  }
  vstar_engine* h = new vstar_engine();
  h->cfg = *cfg;
  if (const char* e = getenv("VSTAR_FUSED_ROPE")) h->fused_rope = atoi(e) != 0;
  if (const char* e = getenv("VSTAR_FOLD_NORMS")) h->fold_norms = atoi(e) != 0;
  if (const char* e = getenv("VSTAR_SCORE_GRAPH")) h->score_graph_on = atoi(e) != 0;
  h->device = device;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&h->stream) != hipSuccess) {
    tls_error() = "hipStreamCreate failed";
    delete h;
    return VSTAR_ERR_HIP;
  }
  if (const char* e = getenv("VSTAR_OWL_OVERLAP")) h->owl_overlap = atoi(e) != 0;
  if (hipStreamCreate(&h->stream2) != hipSuccess || hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess) {
    h->stream2 = nullptr;                       // no second stream: everything stays on the main stream
  }
  // the asynchronous-upload objects exist from the start (created here, on the creating thread; used by the prefetch threads)
  if (hipStreamCreateWithFlags(&h->stream_up, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_stage[0], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_stage[1], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_pre, hipEventDisableTiming) != hipSuccess) {
    tls_error() = "creating the upload stream / events failed";
    vstar_destroy(h);
    return VSTAR_ERR_HIP;
  }
  *out = h;
  return VSTAR_OK;
}

void vstar_destroy(vstar_handle* h) {
  if (!h) return;
  hipSetDevice(h->device);
  hipStreamSynchronize(h->stream);
  vstar_comm_destroy(h);
  h->gen.release();
  h->release_base();
  if (h->d_stats) hipFree(h->d_stats);
  if (h->d_stats_batch) hipFree(h->d_stats_batch);
  if (h->d_up) hipFree(h->d_up);
  for (auto& kv : h->score_graphs) if (kv.second) hipGraphExecDestroy(kv.second);
  h->score_graphs.clear();
  if (h->h_ids_pin) hipHostFree(h->h_ids_pin);
  if (h->h_rowidx_pin) hipHostFree(h->h_rowidx_pin);
  if (h->stream_up) { hipStreamSynchronize(h->stream_up); hipStreamDestroy(h->stream_up); }
  for (auto& ev : h->ev_up) if (ev) hipEventDestroy(ev);
  for (int k = 0; k < 2; ++k) { if (h->ev_stage[k]) hipEventDestroy(h->ev_stage[k]); if (h->h_stage[k]) hipHostFree(h->h_stage[k]); }
  if (h->ev_pre) hipEventDestroy(h->ev_pre);
  for (auto& im : h->images) if (im.d) hipFree(im.d);
  if (h->d_temp) hipFree(h->d_temp);
  if (h->d_tables) hipFree(h->d_tables);
  if (h->stream2) { hipStreamSynchronize(h->stream2); hipStreamDestroy(h->stream2); }
  if (h->ev_fork) hipEventDestroy(h->ev_fork);
  if (h->ev_join) hipEventDestroy(h->ev_join);
  hipStreamDestroy(h->stream);
  delete h;
}

const char* vstar_last_error(const vstar_handle* h) { return h ? h->error.c_str() : tls_error().c_str(); }

int vstar_load_tensor(vstar_handle* h, const char* key, const void* host_ptr, int dtype, int ndim, const int64_t* shape) {
  if (!h || !key || !host_ptr || ndim < 0 || ndim > 8 || (ndim && !shape)) { tls_error() = "bad argument"; return VSTAR_ERR_INVALID; }
  if (h->finalized) { h->set_error("weights already finalized"); return VSTAR_ERR_STATE; }
  return h->stage_tensor(key, host_ptr, dtype, ndim, shape);
}

int vstar_finalize_weights(vstar_handle* h) {
  if (!h) { tls_error() = "null handle"; return VSTAR_ERR_INVALID; }
  return h->finalize();
}

int vstar_vsm_score_batch(vstar_handle* h, int B, const uint16_t* clip_pix, const uint16_t* owl_pix, const int32_t* ids,
                          int L, const int32_t* loc_pos, const int32_t* verify_pos, int n_verify, unsigned flags,
                          vstar_result* out) {
  if (!h) { tls_error() = "null handle"; return VSTAR_ERR_INVALID; }
  return h->score(B, clip_pix, owl_pix, ids, L, loc_pos, verify_pos, n_verify, flags, out);
}

int vstar_vsm_score_grouped(vstar_handle* h, int G, int T, const uint16_t* clip_pix, const uint16_t* owl_pix,
                            const int32_t* prefix_ids, int Lp, const int32_t* suffix_ids, int Ls, const int32_t* loc_in_suffix,
                            const int32_t* verify_in_suffix, int n_verify, unsigned flags, vstar_result* out) {
  if (!h) { tls_error() = "null handle"; return VSTAR_ERR_INVALID; }
  return h->score_grouped(G, T, clip_pix, owl_pix, prefix_ids, Lp, suffix_ids, Ls, loc_in_suffix, verify_in_suffix, n_verify, flags, out);
}

int vstar_vsm_generate(vstar_handle* h, const uint16_t* clip_pix, const int32_t* ids, int L, int max_new_tokens, int eos_id,
                       unsigned flags, int32_t* out_ids, int32_t* n_out) {
  if (!h) { tls_error() = "null handle"; return VSTAR_ERR_INVALID; }
  return h->generate(clip_pix, ids, L, max_new_tokens, eos_id, flags, out_ids, n_out);
}

int vstar_image_set_slot(vstar_handle* h, int slot, const uint8_t* rgb, int height, int width) {
  if (!h || !rgb || height <= 0 || width <= 0 || slot < 0 || slot >= VSTAR_MAX_IMAGE_SLOTS) { tls_error() = "bad argument"; return VSTAR_ERR_INVALID; }
  hipSetDevice(h->device);
  const size_t bytes = (size_t)height * width * 3;
  // crops of the slot's previous image may still be in flight on the stream
  if (hipStreamSynchronize(h->stream) != hipSuccess) { h->set_error("stream sync failed"); return VSTAR_ERR_HIP; }
  if (h->up_pending[slot].exchange(0)) hipEventSynchronize(h->ev_up[slot]);     // an asynchronous upload to this slot still in flight
  auto& im = h->images[slot];
  if (bytes > im.cap) {
    // (d_stats is a fixed-size scratch of vstar_heatmap_stats, independent of the image: it is NOT touched here)
    if (im.d) hipFree(im.d);
    im.d = nullptr;
    im.cap = 0;
    if (hipMalloc((void**)&im.d, bytes) != hipSuccess) { h->set_error("hipMalloc(image) failed"); return VSTAR_ERR_NOMEM; }
    im.cap = bytes;
  }
  if (hipMemcpy(im.d, rgb, bytes, hipMemcpyHostToDevice) != hipSuccess) { h->set_error("image upload failed"); return VSTAR_ERR_HIP; }
  im.H = height;
  im.W = width;
  return VSTAR_OK;
}

int vstar_image_set(vstar_handle* h, const uint8_t* rgb, int height, int width) { return vstar_image_set_slot(h, 0, rgb, height, width); }

int vstar_image_set_slot_async(vstar_handle* h, int slot, const uint8_t* rgb, int height, int width) {
  if (!h || !rgb || height <= 0 || width <= 0 || slot < 0 || slot >= VSTAR_MAX_IMAGE_SLOTS) { tls_error() = "bad argument"; return VSTAR_ERR_INVALID; }
  return h->image_upload_async(slot, rgb, height, width);
}

int vstar_preprocess_crops_slots(vstar_handle* h, int B, const int32_t* boxes_xyxy, const int32_t* slots) {
  if (!h) { tls_error() = "null handle"; return VSTAR_ERR_INVALID; }
  return h->preprocess(B, boxes_xyxy, slots);
}

int vstar_preprocess_crops(vstar_handle* h, int B, const int32_t* boxes_xyxy) { return vstar_preprocess_crops_slots(h, B, boxes_xyxy, nullptr); }

int vstar_heatmap_stats(vstar_handle* h, const float* lowres, int h_out, int w_out, int n_rects, const int32_t* rects_xywh,
                        double* out) {
  if (!h || !lowres || !out || h_out <= 0 || w_out <= 0 || n_rects < 0 || n_rects > 8 || (n_rects && !rects_xywh)) {
    tls_error() = "bad argument";
    return VSTAR_ERR_INVALID;
  }
  // the one-map form of vstar_heatmap_stats_batch: same kernel, same launch geometry per map, same numbers
  int32_t hw[2] = {h_out, w_out}, rects[32] = {0};
  if (n_rects) memcpy(rects, rects_xywh, (size_t)n_rects * 16);
  double all[11];
  const int rc = vstar_heatmap_stats_batch(h, 1, lowres, hw, &n_rects, rects, all);
  if (rc != VSTAR_OK) return rc;
  memcpy(out, all, sizeof(double) * (3 + n_rects));
  return VSTAR_OK;
}

int vstar_heatmap_stats_batch(vstar_handle* h, int n, const float* lowres, const int32_t* out_hw, const int32_t* n_rects,
                              const int32_t* rects_xywh, double* out) {
  if (!h || n < 0 || (n && (!lowres || !out_hw || !n_rects || !rects_xywh || !out))) { tls_error() = "bad argument"; return VSTAR_ERR_INVALID; }
  if (n == 0) return VSTAR_OK;
  int64_t max_pixels = 0;
  for (int i = 0; i < n; ++i) {
    if (out_hw[2 * i] <= 0 || out_hw[2 * i + 1] <= 0 || n_rects[i] < 0 || n_rects[i] > 8) { tls_error() = "bad argument"; return VSTAR_ERR_INVALID; }
    const int64_t px = (int64_t)out_hw[2 * i] * out_hw[2 * i + 1];
    max_pixels = px > max_pixels ? px : max_pixels;
  }
  hipSetDevice(h->device);
  constexpr size_t MAPF = (size_t)VSTAR_MASK_RES * VSTAR_MASK_RES;
  // device scratch per item: [low MAPF f32][out 11 f64][rects 32 i32][mm 2 u32][hw 2 i32][n_rects 1 i32 (+1 pad)]; grown on demand.
  // ONE launch scores all n maps (heat_stats_batch; round 3 queued three launches per map)
  const size_t per = MAPF * 4 + 11 * 8 + 32 * 4 + 2 * 4 + 2 * 4 + 2 * 4;
  const size_t need = per * (size_t)n;
  if (need > h->stats_batch_cap) {
    if (hipStreamSynchronize(h->stream) != hipSuccess) { h->set_error("stream sync failed"); return VSTAR_ERR_HIP; }
    if (h->d_stats_batch) hipFree(h->d_stats_batch);
    h->d_stats_batch = nullptr; h->stats_batch_cap = 0;
    const size_t want = need + need / 2;
    if (hipMalloc(&h->d_stats_batch, want) != hipSuccess) { h->set_error("hipMalloc failed in vstar_heatmap_stats_batch"); return VSTAR_ERR_NOMEM; }
    h->stats_batch_cap = want;
  }
  char* base = (char*)h->d_stats_batch;
  float* d_low = (float*)base;
  double* d_out = (double*)(base + MAPF * 4 * n);
  int* d_rects = (int*)(base + (MAPF * 4 + 11 * 8) * n);
  unsigned* d_mm = (unsigned*)(base + (MAPF * 4 + 11 * 8 + 32 * 4) * n);
  int* d_hw = (int*)(base + (MAPF * 4 + 11 * 8 + 32 * 4 + 2 * 4) * n);
  int* d_nr = (int*)(base + (MAPF * 4 + 11 * 8 + 32 * 4 + 2 * 4 + 2 * 4) * n);
  bool ok = hipMemcpyAsync(d_low, lowres, MAPF * 4 * n, hipMemcpyHostToDevice, h->stream) == hipSuccess;
  ok = ok && hipMemcpyAsync(d_rects, rects_xywh, (size_t)32 * 4 * n, hipMemcpyHostToDevice, h->stream) == hipSuccess;
  ok = ok && hipMemcpyAsync(d_hw, out_hw, (size_t)2 * 4 * n, hipMemcpyHostToDevice, h->stream) == hipSuccess;
  ok = ok && hipMemcpyAsync(d_nr, n_rects, (size_t)4 * n, hipMemcpyHostToDevice, h->stream) == hipSuccess;
  ok = ok && heat_stats_batch(d_low, VSTAR_MASK_RES, VSTAR_MASK_RES, d_hw, d_rects, d_nr, n, max_pixels, d_out, d_mm, h->stream) == hipSuccess;
  ok = ok && hipMemcpyAsync(out, d_out, sizeof(double) * 11 * n, hipMemcpyDeviceToHost, h->stream) == hipSuccess;
  ok = ok && hipStreamSynchronize(h->stream) == hipSuccess;
  if (!ok) { h->set_error("vstar_heatmap_stats_batch: HIP failure"); return VSTAR_ERR_HIP; }
  return VSTAR_OK;
}

int vstar_upsample_mask(vstar_handle* h, const float* lowres, int h_out, int w_out, float* out) {
  return vstar_upsample_mask_ex(h, lowres, h_out, w_out, 1, out);
}

int vstar_upsample_mask_ex(vstar_handle* h, const float* lowres, int h_out, int w_out, int clamp_min0, float* out) {
  if (!h || !lowres || !out || h_out <= 0 || w_out <= 0) { tls_error() = "bad argument"; return VSTAR_ERR_INVALID; }
  hipSetDevice(h->device);
  // persistent grow-only scratch (round 2 allocated and freed two buffers per call: the cue / segmentation path and every
  // committed node of a host-reductions search paid two hipMalloc/hipFree pairs)
  const size_t nin = (size_t)VSTAR_MASK_RES * VSTAR_MASK_RES, nout = (size_t)h_out * w_out;
  if (nin + nout > h->up_cap) {
    if (hipStreamSynchronize(h->stream) != hipSuccess) { h->set_error("stream sync failed"); return VSTAR_ERR_HIP; }
    if (h->d_up) hipFree(h->d_up);
    h->d_up = nullptr; h->up_cap = 0;
    const size_t want = nin + nout + nout / 4;
    if (hipMalloc((void**)&h->d_up, want * 4) != hipSuccess) { h->set_error("hipMalloc failed in vstar_upsample_mask"); return VSTAR_ERR_NOMEM; }
    h->up_cap = want;
  }
  float *din = h->d_up, *dout = h->d_up + nin;
  hipError_t e;
  const char* what = "H2D copy";
  if ((e = hipMemcpyAsync(din, lowres, nin * 4, hipMemcpyHostToDevice, h->stream)) == hipSuccess) {
    what = "resize kernel";
    if ((e = resize_bilinear_clamp(din, VSTAR_MASK_RES, VSTAR_MASK_RES, dout, h_out, w_out, h->stream, clamp_min0)) == hipSuccess) {
      what = "D2H copy";
      if ((e = hipMemcpyAsync(out, dout, nout * 4, hipMemcpyDeviceToHost, h->stream)) == hipSuccess) {
        what = "stream sync";
        e = hipStreamSynchronize(h->stream);
      }
    }
  }
  if (e != hipSuccess) {
    char dbg[256];
    hipPointerAttribute_t at{};
    const hipError_t pa = hipPointerGetAttributes(&at, din);
    snprintf(dbg, sizeof(dbg), " [din=%p cap=%zu nin=%zu nout=%zu lowres=%p out=%p attr=%d type=%d dev=%d]", (void*)din, h->up_cap, nin, nout,
             (const void*)lowres, (void*)out, (int)pa, (int)at.type, at.device);
    h->set_error(std::string("vstar_upsample_mask: ") + what + ": " + hipGetErrorString(e) + dbg);
    return VSTAR_ERR_HIP;
  }
  return VSTAR_OK;
}

int64_t vstar_debug_read(vstar_handle* h, const char* name, float* out, int64_t cap) {
  if (!h || !name || !out) { tls_error() = "bad argument"; return VSTAR_ERR_INVALID; }
  if (!h->finalized) { h->set_error("weights not finalized"); return VSTAR_ERR_STATE; }
  hipSetDevice(h->device);
  if (!strcmp(name, "score_graph_stats")) {         // host counters: [replays of a captured scoring graph, captured graphs, signatures seen]
    if (cap < 3) { h->set_error("score_graph_stats needs room for 3 values"); return VSTAR_ERR_INVALID; }
    int captured = 0;
    for (auto& kv : h->score_graphs) captured += kv.second != nullptr;
    out[0] = (float)h->score_graph_replays; out[1] = (float)captured; out[2] = (float)h->score_graphs.size();
    return 3;
  }
  const bool pix = !strcmp(name, "clip_pixels") || !strcmp(name, "owl_pixels");
  if (!pix && h->last_B == 0) { h->set_error("no forward has run"); return VSTAR_ERR_STATE; }
  const int B = pix ? (int)h->h_jobs.size() / 2 : h->last_B;
  const std::string n(name);
  const lp_t* src = nullptr;
  int64_t cnt = 0;
  if (n == "clip_pixels") { src = h->d_clip_pix; cnt = (int64_t)B * 3 * h->cfg.clip_image_size * h->cfg.clip_image_size; }
  else if (n == "owl_pixels") { src = h->d_owl_pix; cnt = (int64_t)B * 3 * h->cfg.owl_image_size * h->cfg.owl_image_size; }
  else if (n == "clip_features") { src = h->clip.x; cnt = (int64_t)B * h->clip.N * h->clip.hidden; }
  else if (n == "llm_input") { src = h->lx; cnt = 0; }
  else if (n == "llm_hidden_loc") { src = h->hsel; cnt = (int64_t)B * h->cfg.llm_hidden; }
  else if (n == "embed_det") { src = h->emb_det; cnt = (int64_t)B * h->det1.N; }
  else if (n == "embed_seg") { src = h->emb_seg; cnt = (int64_t)B * 256; }
  else if (n == "owl_feats") { src = h->owl_feats; cnt = (int64_t)B * h->owl.P * h->owl.hidden; }
  else if (n == "sam_tokens") { src = h->s_q; cnt = (int64_t)B * 6 * 256; }
  else if (n == "sam_keys") { src = h->s_keys; cnt = (int64_t)B * h->owl.P * 256; }
  else if (n == "sam_src") { src = h->s_src; cnt = (int64_t)B * h->owl.P * 256; }
  else if (n == "sam_c1") { src = h->s_c1; cnt = (int64_t)B * 96 * 96 * 64; }       // Upsample(256->64) output, channels-last
  else if (n == "sam_c1n") { src = h->s_c1n; cnt = (int64_t)B * 96 * 96 * 64; }     // LayerNorm2d + GELU
  else if (n == "sam_c2") { src = h->s_c2; cnt = (int64_t)B * 192 * 192 * 32; }     // Upsample(64->32) + GELU
  else if (n == "sam_hyper") { src = h->s_hyper; cnt = (int64_t)B * 32; }
  else { h->set_error("unknown debug tensor: " + n); return VSTAR_ERR_INVALID; }
  if (cnt > cap) cnt = cap;
  std::vector<lp_t> tmp((size_t)cnt);
  if (hipStreamSynchronize(h->stream) != hipSuccess ||
      hipMemcpy(tmp.data(), src, (size_t)cnt * 2, hipMemcpyDeviceToHost) != hipSuccess) {
    h->set_error("debug_read: HIP failure");
    return VSTAR_ERR_HIP;
  }
  for (int64_t i = 0; i < cnt; ++i) out[i] = lp2f(tmp[(size_t)i]);
  return cnt;
}

void* vstar_stream(vstar_handle* h) { return h ? (void*)h->stream : nullptr; }

int vstar_profile_enable(vstar_handle* h, int on) {
  if (!h) return VSTAR_ERR_INVALID;
  h->profile = on != 0;
  h->prof_ms = 0; h->prof_flops = 0; h->prof_launches = 0; h->pending_flops = 0; h->ev_used = 0;
  h->prof_ms8 = 0; h->prof_flops8 = 0; h->prof_launches8 = 0; h->ev_flops.clear();
  return VSTAR_OK;
}
int vstar_profile_read(vstar_handle* h, double* gemm_ms, int64_t* gemm_launches, double* gemm_flops) {
  if (!h) return VSTAR_ERR_INVALID;
  hipSetDevice(h->device);
  hipStreamSynchronize(h->stream);
  h->collect_profile();
  if (gemm_ms) *gemm_ms = h->prof_ms;
  if (gemm_launches) *gemm_launches = h->prof_launches;
  if (gemm_flops) *gemm_flops = h->prof_flops;
  return VSTAR_OK;
}

int vstar_profile_read_fp8(vstar_handle* h, double* gemm_ms, int64_t* gemm_launches, double* gemm_flops) {
  if (!h) return VSTAR_ERR_INVALID;
  hipSetDevice(h->device);
  hipStreamSynchronize(h->stream);
  h->collect_profile();
  if (gemm_ms) *gemm_ms = h->prof_ms8;
  if (gemm_launches) *gemm_launches = h->prof_launches8;
  if (gemm_flops) *gemm_flops = h->prof_flops8;
  return VSTAR_OK;
}

// ---------------- operator-level entry points ----------------
static int op_rc(hipError_t e) {
  if (e != hipSuccess) { tls_error() = std::string("HIP: ") + hipGetErrorString(e); return VSTAR_ERR_HIP; }
  return VSTAR_OK;
}

int vstar_op_gemm(void* stream, const uint16_t* A, int64_t lda, const uint16_t* W, const uint16_t* bias,
                  const uint16_t* residual, int64_t ldr, void* C, int64_t ldc, int out_f32, int M, int N, int K,
                  int epilogue) {
  GemmParams p{};
  p.A = A; p.lda = lda; p.W = W; p.bias = bias; p.res = residual; p.ldr = ldr; p.C = C; p.ldc = ldc;
  p.M = M; p.N = N; p.K = K;
  { const char* e = getenv("VSTAR_GEMM_DEBUG"); p.debug_flags = e ? atoi(e) : 0; }
  const bool nosync = (epilogue & VSTAR_EPI_NOSYNC) != 0;
  if ((epilogue & VSTAR_EPI_TILE128) && (epilogue & VSTAR_EPI_TILE256)) { tls_error() = "both tile overrides set"; return VSTAR_ERR_INVALID; }
  p.tile_force = (epilogue & VSTAR_EPI_TILE4W) ? GEMM_TILE_4W : (epilogue & VSTAR_EPI_TILE256) ? 256 : (epilogue & VSTAR_EPI_TILE128) ? 128 : 0;
  if (p.tile_force == 256 && !gemm256_eligible(p)) {
    tls_error() = "VSTAR_EPI_TILE256: shape outside the 256x256 kernel's domain (M >= 1024, N >= 256, K % 128 == 0)";
    return VSTAR_ERR_INVALID;
  }
  hipError_t e = gemm_lp(p, epilogue & 0xff, out_f32 != 0, (hipStream_t)stream);
  if (e == hipSuccess && !nosync) e = hipStreamSynchronize((hipStream_t)stream);
  return op_rc(e);
}
int vstar_op_gemm_norm(void* stream, const uint16_t* A, int64_t lda, const uint16_t* W, const uint16_t* bias,
                       const uint16_t* residual, int64_t ldr, void* C, int64_t ldc, int M, int N, int K, int epilogue,
                       const float* row_scale, float* sumsq_out, int sumsq_ld) {
  GemmParams p{};
  p.A = A; p.lda = lda; p.W = W; p.bias = bias; p.res = residual; p.ldr = ldr; p.C = C; p.ldc = ldc;
  p.M = M; p.N = N; p.K = K;
  p.row_scale = row_scale; p.sumsq_out = sumsq_out; p.sumsq_ld = sumsq_ld;
  { const char* e = getenv("VSTAR_GEMM_DEBUG"); p.debug_flags = e ? atoi(e) : 0; }
  if ((epilogue & VSTAR_EPI_TILE128) && (epilogue & VSTAR_EPI_TILE256)) { tls_error() = "both tile overrides set"; return VSTAR_ERR_INVALID; }
  if (sumsq_out && ((epilogue & 0xff) != VSTAR_EPI_NONE || N % 64)) { tls_error() = "sumsq_out: VSTAR_EPI_NONE and N % 64 == 0 only"; return VSTAR_ERR_INVALID; }
  p.tile_force = (epilogue & VSTAR_EPI_TILE4W) ? GEMM_TILE_4W : (epilogue & VSTAR_EPI_TILE256) ? 256 : (epilogue & VSTAR_EPI_TILE128) ? 128 : 0;
  if (p.tile_force == 256 && !gemm256_eligible(p)) { tls_error() = "VSTAR_EPI_TILE256: shape outside the 256x256 kernel's domain"; return VSTAR_ERR_INVALID; }
  hipError_t e = gemm_lp(p, epilogue & 0xff, false, (hipStream_t)stream);
  if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
  return op_rc(e);
}
int vstar_op_rms_rstd(void* stream, const uint16_t* x, const float* partials, int ld, int rows, int cols, float eps, float* r) {
  hipError_t e = x ? rms_rstd_rows(x, rows, cols, eps, r, (hipStream_t)stream)
                   : rms_rstd_partials(partials, ld, rows, cols, eps, r, (hipStream_t)stream);
  if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
  return op_rc(e);
}
int vstar_op_ln_fold(void* stream, uint16_t* W, uint16_t* bias, const uint16_t* g, const uint16_t* b_ln, int n_rows, int K) {
  hipError_t e = ln_fold_weights(W, bias, g, b_ln, n_rows, K, (hipStream_t)stream);
  if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
  return op_rc(e);
}
int vstar_op_gemm_last_tile(void) { return gemm_last_tile(); }
int vstar_op_gemm_plan(int M, int N, int K, int epilogue, int has_residual, int fused_rope, int cus) {
  if (M <= 0 || N <= 0 || K <= 0 || cus <= 0) { tls_error() = "vstar_op_gemm_plan: bad argument"; return VSTAR_ERR_INVALID; }
  // nothing is dereferenced in a dry run: the pointers only have to look aligned and non-null
  static const uintptr_t fake = 0x10000;
  GemmParams p{};
  p.A = (const lp_t*)fake; p.lda = K; p.W = (const lp_t*)fake; p.C = (void*)fake;
  p.M = M; p.N = N; p.K = K;
  const int n_out = (epilogue & 0xff) == VSTAR_EPI_SILU_MUL ? N / 2 : N;
  p.ldc = n_out;
  if (has_residual) { p.res = (const lp_t*)fake; p.ldr = n_out; }
  if (fused_rope) { p.rope_cs = (const lp_t*)fake; p.rope_S = 640; p.rope_cols = N * 2 / 3; }
  p.tile_force = (epilogue & VSTAR_EPI_TILE256) ? 256 : (epilogue & VSTAR_EPI_TILE128) ? 128 : 0;
  gemm_set_plan(true, cus);
  const hipError_t e = gemm_lp(p, epilogue & 0xff, false, nullptr);
  gemm_set_plan(false, 0);
  if (e != hipSuccess) { tls_error() = std::string("vstar_op_gemm_plan: ") + hipGetErrorString(e); return VSTAR_ERR_INVALID; }
  return gemm_last_tile() * 10 + gemm_last_mode();
}
int vstar_op_layernorm(void* stream, const uint16_t* x, const uint16_t* g, const uint16_t* b, uint16_t* y, int rows,
                       int cols, float eps) {
  hipError_t e = layernorm_lp(x, g, b, y, rows, cols, eps, nullptr, 0, (hipStream_t)stream);
  if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
  return op_rc(e);
}
int vstar_op_rmsnorm(void* stream, const uint16_t* x, const uint16_t* g, uint16_t* y, int rows, int cols, float eps) {
  hipError_t e = rmsnorm_lp(x, g, y, rows, cols, eps, nullptr, (hipStream_t)stream);
  if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
  return op_rc(e);
}
int vstar_op_gemm_fp8(void* stream, const uint16_t* A, const uint16_t* W, const uint16_t* bias, const uint16_t* res, uint16_t* C,
                      int M, int N, int K, int epilogue, int iters, float* gemm_ms) {
  if (!A || !W || !C || M <= 0 || N <= 0 || K <= 0) { tls_error() = "bad argument"; return VSTAR_ERR_INVALID; }
  hipStream_t s = (hipStream_t)stream;
  const int Npad = (N + 255) / 256 * 256;
  const int n_out = (epilogue & 0xff) == VSTAR_EPI_SILU_MUL ? N / 2 : N;
  uint8_t *Aq = nullptr, *Wq = nullptr;
  float *sa = nullptr, *sw = nullptr;
  hipError_t e = hipMalloc((void**)&Aq, (size_t)M * K);
  if (e == hipSuccess) e = hipMalloc((void**)&Wq, (size_t)Npad * K);
  if (e == hipSuccess) e = hipMalloc((void**)&sa, (size_t)M * 4);
  if (e == hipSuccess) e = hipMalloc((void**)&sw, (size_t)Npad * 4);
  if (e == hipSuccess) e = quantize_rows_fp8(A, K, Aq, K, sa, M, K, s);
  if (e == hipSuccess) e = quantize_rows_fp8(W, K, Wq, K, sw, Npad, K, s);
  GemmParams p{};
  p.A = (const lp_t*)Aq; p.lda = K; p.W = (const lp_t*)Wq; p.bias = bias; p.res = res; p.ldr = n_out; p.C = C; p.ldc = n_out;
  p.M = M; p.N = N; p.K = K; p.a_scale = sa; p.w_scale = sw;
  // VSTAR_EPI_TILE256 / VSTAR_EPI_TILE4W OR-ed into `epilogue`: force the 8-wave / the 4-wave W8A8 kernel (round 6; bit-identity tests)
  p.tile_force = (epilogue & VSTAR_EPI_TILE4W) ? GEMM_TILE_4W : (epilogue & VSTAR_EPI_TILE256) ? 256 : 0;
  epilogue &= 0xff;
  if (e == hipSuccess && !gemm256_eligible(p)) { tls_error() = "shape not accepted by the W8A8 kernel"; e = hipErrorInvalidValue; }
  if (e == hipSuccess) e = gemm_lp(p, epilogue, false, s);
  if (e == hipSuccess && iters > 0 && gemm_ms) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, s);
    for (int i = 0; i < iters && e == hipSuccess; ++i) e = gemm_lp(p, epilogue, false, s);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    *gemm_ms = ms / iters;
    hipEventDestroy(e0); hipEventDestroy(e1);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  hipFree(Aq); hipFree(Wq); hipFree(sa); hipFree(sw);
  return op_rc(e);
}
// ---- block-scaled W8A8 (mx.hpp), op level ----
size_t vstar_op_mx_scale_bytes(int rows, int cols) { return (rows % 128 || cols % 128) ? 0 : mx_scale_bytes(rows, cols); }
int64_t vstar_op_mx_scale_offset(int row, int k_block, int rows) { return mx_scale_offset(row, k_block, rows >> 7); }
int vstar_op_quantize_mx(void* stream, const uint16_t* X, uint8_t* q, uint8_t* scales, int rows, int cols) {
  hipError_t e = quantize_rows_mx(X, cols, q, cols, scales, rows, cols, (hipStream_t)stream);
  if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
  return op_rc(e);
}
int vstar_op_gemm_mx(void* stream, const uint8_t* Aq, const uint8_t* a_scales, const float* row_scale, const uint16_t* W, const uint16_t* res,
                     uint16_t* C, uint8_t* C8, uint8_t* c_scales, float* sumsq, int M, int N, int K, int epilogue, int iters, float* gemm_ms) {
  if (!Aq || !a_scales || !W || (!C && !C8) || M <= 0 || N <= 0 || K <= 0) { tls_error() = "bad argument"; return VSTAR_ERR_INVALID; }
  hipStream_t s = (hipStream_t)stream;
  const int Npad = (N + 255) / 256 * 256;
  const int n_out = epilogue == VSTAR_EPI_SILU_MUL ? N / 2 : N;
  uint8_t* Wq = nullptr;
  float* sw = nullptr;
  hipError_t e = hipMalloc((void**)&Wq, (size_t)Npad * K);
  if (e == hipSuccess) e = hipMalloc((void**)&sw, (size_t)Npad * 4);
  if (e == hipSuccess) e = quantize_rows_fp8(W, K, Wq, K, sw, Npad, K, s);
  GemmParams p{};
  p.A = (const lp_t*)Aq; p.lda = K; p.W = (const lp_t*)Wq; p.res = res; p.ldr = n_out; p.M = M; p.N = N; p.K = K;
  p.w_scale = sw; p.a_mx = a_scales; p.a_scale = row_scale;
  if (epilogue == VSTAR_EPI_SILU_MUL && C8) { p.C = C8; p.ldc = n_out; p.c_mx = c_scales; }       // fp8 out only
  else { p.C = C; p.ldc = n_out; p.c8 = C8; p.ldc8 = n_out; p.c_mx = C8 ? c_scales : nullptr; p.sumsq_out = C8 ? sumsq : nullptr; p.sumsq_ld = n_out / 64; }
  if (e == hipSuccess && !gemm_mx_supported(p, epilogue)) { tls_error() = "shape not accepted by the block-scaled W8A8 kernel"; e = hipErrorInvalidValue; }
  if (e == hipSuccess) e = gemm_lp(p, epilogue, false, s);
  if (e == hipSuccess && iters > 0 && gemm_ms) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, s);
    for (int i = 0; i < iters && e == hipSuccess; ++i) e = gemm_lp(p, epilogue, false, s);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    *gemm_ms = ms / iters;
    hipEventDestroy(e0); hipEventDestroy(e1);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  hipFree(Wq); hipFree(sw);
  return op_rc(e);
}
int vstar_op_gemm_fp8_mxout(void* stream, const uint16_t* A, const uint16_t* W, uint8_t* C8, uint8_t* c_scales, int M, int N, int K, int iters,
                            float* gemm_ms) {
  if (!A || !W || !C8 || !c_scales || M <= 0 || N <= 0 || K <= 0) { tls_error() = "bad argument"; return VSTAR_ERR_INVALID; }
  hipStream_t s = (hipStream_t)stream;
  const int Npad = (N + 255) / 256 * 256;
  uint8_t *Aq = nullptr, *Wq = nullptr;
  float *sa = nullptr, *sw = nullptr;
  hipError_t e = hipMalloc((void**)&Aq, (size_t)M * K);
  if (e == hipSuccess) e = hipMalloc((void**)&Wq, (size_t)Npad * K);
  if (e == hipSuccess) e = hipMalloc((void**)&sa, (size_t)M * 4);
  if (e == hipSuccess) e = hipMalloc((void**)&sw, (size_t)Npad * 4);
  if (e == hipSuccess) e = quantize_rows_fp8(A, K, Aq, K, sa, M, K, s);
  if (e == hipSuccess) e = quantize_rows_fp8(W, K, Wq, K, sw, Npad, K, s);
  GemmParams p{};
  p.A = (const lp_t*)Aq; p.lda = K; p.W = (const lp_t*)Wq; p.C = C8; p.ldc = N / 2;
  p.M = M; p.N = N; p.K = K; p.a_scale = sa; p.w_scale = sw; p.c_mx = c_scales;
  if (e == hipSuccess && !gemm_mx_supported(p, VSTAR_EPI_SILU_MUL)) { tls_error() = "shape not accepted by the block-scaled W8A8 kernel"; e = hipErrorInvalidValue; }
  if (e == hipSuccess) e = gemm_lp(p, VSTAR_EPI_SILU_MUL, false, s);
  if (e == hipSuccess && iters > 0 && gemm_ms) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, s);
    for (int i = 0; i < iters && e == hipSuccess; ++i) e = gemm_lp(p, VSTAR_EPI_SILU_MUL, false, s);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    *gemm_ms = ms / iters;
    hipEventDestroy(e0); hipEventDestroy(e1);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  hipFree(Aq); hipFree(Wq); hipFree(sa); hipFree(sw);
  return op_rc(e);
}
int vstar_op_attention_mx(void* stream, const uint16_t* qkv, uint8_t* out8, uint8_t* scales, int B, int S, int H) {
  hipError_t e = attn_forward_mx(qkv, out8, scales, B, S, H, 1.0f / sqrtf(128.0f), (hipStream_t)stream);
  if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
  return op_rc(e);
}
int vstar_w8a8_mx_active(vstar_handle* h) { return h ? h->last_w8a8_chain : VSTAR_ERR_INVALID; }

size_t vstar_op_attention_workspace(int B, int S, int H, int D) {
  (void)B; (void)H;
  return (size_t)S * D * 2 + 256;     // the RoPE cos|sin table only: V is transposed inside the kernel (no V^T buffer)
}
int vstar_op_attention(void* stream, uint16_t* qkv, uint16_t* out, void* ws, size_t ws_bytes, int B, int S, int H, int D,
                       int causal, float rope_theta) {
  if (ws_bytes < vstar_op_attention_workspace(B, S, H, D)) { tls_error() = "attention workspace too small"; return VSTAR_ERR_INVALID; }
  hipStream_t s = (hipStream_t)stream;
  lp_t* cs = nullptr;
  if (rope_theta > 0.f) {
    cs = (lp_t*)ws;
    std::vector<lp_t> tab((size_t)S * D);
    for (int p = 0; p < S; ++p)
      for (int i = 0; i < D / 2; ++i) {
        const float inv = 1.0f / powf(rope_theta, (float)(2 * i) / (float)D);
        tab[(size_t)p * D + i] = f2lp(cosf((float)p * inv));
        tab[(size_t)p * D + D / 2 + i] = f2lp(sinf((float)p * inv));
      }
    hipError_t e = hipMemcpy(cs, tab.data(), tab.size() * 2, hipMemcpyHostToDevice);
    if (e != hipSuccess) return op_rc(e);
  }
  hipError_t e = attn_prepare(qkv, cs, B, S, H, D, s);
  if (e == hipSuccess) e = attn_forward(qkv, out, B, S, H, D, causal, 1.0f / sqrtf((float)D), s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  return op_rc(e);
}

}  // extern "C"

// accessors for comm.hip (the handle's layout is private to this file)
int vstar_handle_device(vstar_handle* h) { return h->device; }
void vstar_handle_set_error(vstar_handle* h, const char* msg) { h->set_error(msg); }
void** vstar_handle_comm_slot(vstar_handle* h) { return &h->comm; }
