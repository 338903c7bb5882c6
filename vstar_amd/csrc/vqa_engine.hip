// vqa_engine.hip — host side of the VQA-LLM engine (include/vstar_vqa.h); compiled with -DVSTAR_LP_F16 only.
// Mirrors LlavaSearchLlamaForCausalLM (LLaVA/llava/model/language_model/llava_search_llama.py:40-113,
// LLaVA/llava/model/llava_search_arch.py:84-266, multimodal_projector/{builder,perceiver}.py) as used by
// VQA_LLM.{free_form_inference,multiple_choices_inference} (vstar_bench_eval.py:78-165).
//
// The KV-cached language-model forward itself is llm_cached.hpp (shared with the VSM engine's free-text decode); this file
// owns the weights, the vision side (CLIP tower, mm_projector, Perceiver object projector) and the feature table.
#ifndef VSTAR_LP_F16
#error "vqa_engine.hip is the fp16 instantiation: build with -DVSTAR_LP_F16"
#endif
#include "llm_cached.hpp"
#include "../../include/vstar_vqa.h"


namespace {
struct PcvLayer { lp_t *nm_g, *nm_b, *nl_g, *nl_b, *ff_g, *ff_b; Lin to_q, to_kv, to_out, ff1, ff2; };
}  // namespace

struct vstar_vqa_engine : EngineBase {
  vstar_vqa_config cfg{};
  VitTower clip;
  Lin proj0, proj1;                 // mm_projector (proj1 only for mlp2x_gelu)
  lp_t *pcv_ln_g = nullptr, *pcv_ln_b = nullptr, *pcv_latents = nullptr, *pcv_media_pos = nullptr, *pcv_norm_g = nullptr,
       *pcv_norm_b = nullptr;
  std::vector<PcvLayer> pcv;
  Lin pcv_out;
  lp_t* embed = nullptr;
  std::vector<LlmBlock> llm;
  lp_t* final_norm = nullptr;
  Lin lm_head;
  LlmCached run;                    // KV cache + the language-model forward (llm_cached.hpp)
  lp_t* feats = nullptr;            // feature table [max_images * (P + L), H]
  int32_t *d_latidx = nullptr, *d_patchidx = nullptr;
  lp_t* d_pix = nullptr;
  // perceiver activations
  lp_t *p_xm = nullptr, *p_nm = nullptr, *p_lat = nullptr, *p_nl = nullptr, *p_q = nullptr, *p_kv = nullptr, *p_att = nullptr,
       *p_ff = nullptr, *p_tmp = nullptr;
  int enc_batch = 0;

  int finalize();
  int encode(int n, const uint16_t* pix, int first_slot);
};

int vstar_vqa_engine::finalize() {
  if (finalized) { set_error("weights already finalized"); return VSTAR_ERR_STATE; }
  const vstar_vqa_config& c = cfg;
  HIPCHK(hipSetDevice(device));
  const int clip_blocks = c.clip_layers + 1 + c.clip_select_layer;
  if (clip_blocks < 0 || clip_blocks > c.clip_layers) { set_error("bad clip_select_layer"); return VSTAR_ERR_INVALID; }
  enc_batch = c.max_images < 8 ? c.max_images : 8;
  RC(build_tower(clip, "clip.vision_model.", "pre_layrnorm", c.clip_image_size, c.clip_patch, c.clip_hidden, c.clip_heads,
                 c.clip_mlp, clip_blocks, enc_batch));
  const int C = c.clip_hidden, H = c.llm_hidden, P = clip.P, L = c.pcv_latents;
  if (c.llm_mlp % 16) { set_error("llm_mlp must be a multiple of 16"); return VSTAR_ERR_INVALID; }
  // ---- mm_projector (builder.py:39-49) ----
  if (c.projector_type == 0) {
    RC(make_lin({"model.mm_projector.weight"}, {"model.mm_projector.bias"}, &proj0, C));
  } else if (c.projector_type == 1) {
    RC(make_lin({"model.mm_projector.0.weight"}, {"model.mm_projector.0.bias"}, &proj0, C));
    RC(make_lin({"model.mm_projector.2.weight"}, {"model.mm_projector.2.bias"}, &proj1, H));
  } else { set_error("unknown projector_type"); return VSTAR_ERR_INVALID; }
  // ---- mm_projector_object = Sequential(LayerNorm, PerceiverResampler, Linear) (builder.py:54-66) ----
  const std::string po = "model.mm_projector_object.";
  const int inner = c.pcv_heads * c.pcv_dim_head;
  RC(upload_vec(po + "0.weight", &pcv_ln_g, C));
  RC(upload_vec(po + "0.bias", &pcv_ln_b, C));
  RC(upload_vec(po + "1.latents", &pcv_latents, (int64_t)L * C));
  RC(upload_vec(po + "1.media_pos_emb", &pcv_media_pos, C));     // [1,1,C]: num_media_embeds = 1
  pcv.resize(c.pcv_depth);
  for (int i = 0; i < c.pcv_depth; ++i) {
    const std::string lp = po + "1.layers." + std::to_string(i) + ".";
    PcvLayer& l = pcv[i];
    RC(upload_vec(lp + "0.norm_media.weight", &l.nm_g, C));
    RC(upload_vec(lp + "0.norm_media.bias", &l.nm_b, C));
    RC(upload_vec(lp + "0.norm_latents.weight", &l.nl_g, C));
    RC(upload_vec(lp + "0.norm_latents.bias", &l.nl_b, C));
    RC(make_lin({lp + "0.to_q.weight"}, {}, &l.to_q, C));
    RC(make_lin({lp + "0.to_kv.weight"}, {}, &l.to_kv, C));
    RC(make_lin({lp + "0.to_out.weight"}, {}, &l.to_out, inner));
    RC(upload_vec(lp + "1.0.weight", &l.ff_g, C));
    RC(upload_vec(lp + "1.0.bias", &l.ff_b, C));
    RC(make_lin({lp + "1.1.weight"}, {}, &l.ff1, C));
    RC(make_lin({lp + "1.3.weight"}, {}, &l.ff2, C * c.pcv_ff_mult));
    if (l.to_q.N != inner || l.to_kv.N != 2 * inner) { set_error("perceiver head geometry mismatch"); return VSTAR_ERR_INVALID; }
  }
  RC(upload_vec(po + "1.norm.weight", &pcv_norm_g, C));
  RC(upload_vec(po + "1.norm.bias", &pcv_norm_b, C));
  RC(make_lin({po + "2.weight"}, {po + "2.bias"}, &pcv_out, C));
  // ---- LLaMA ----
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
  RC(upload_vec("model.norm.weight", &final_norm, H));
  RC(make_lin({"lm_head.weight"}, {}, &lm_head, H));
  // ---- feature table, language-model runner (KV cache, activations) ----
  RC(dalloc(&feats, (size_t)c.max_images * (P + L) * H));
  {
    LlmCachedCfg rc;
    rc.hidden = H; rc.heads = c.llm_heads; rc.mlp = c.llm_mlp; rc.layers = c.llm_layers; rc.vocab = c.llm_vocab;
    rc.rms_eps = c.llm_rms_eps; rc.rope_theta = c.llm_rope_theta;
    rc.max_slots = c.max_slots; rc.max_ctx = c.max_ctx; rc.max_rows = c.max_rows;
    RC(run.init(this, rc, embed, &llm, final_norm, &lm_head));
    run.feats = feats;
    run.n_feat_rows = (int64_t)c.max_images * (P + L);
  }
  const int nb = enc_batch;
  RC(dalloc(&d_pix, (size_t)nb * 3 * c.clip_image_size * c.clip_image_size));
  RC(dalloc(&p_xm, (size_t)nb * P * C));
  RC(dalloc(&p_nm, (size_t)nb * P * C));
  RC(dalloc(&p_lat, (size_t)nb * L * C));
  RC(dalloc(&p_nl, (size_t)nb * L * C));
  RC(dalloc(&p_q, (size_t)nb * L * inner));
  RC(dalloc(&p_kv, (size_t)nb * (P + L) * 2 * inner));
  RC(dalloc(&p_att, (size_t)nb * L * inner));
  RC(dalloc(&p_ff, (size_t)nb * L * C * c.pcv_ff_mult));
  RC(dalloc(&p_tmp, (size_t)nb * P * H));
  {
    std::vector<int32_t> li((size_t)nb * L), pi((size_t)nb * P);
    for (int i = 0; i < nb * L; ++i) li[i] = i % L;
    for (int i = 0; i < nb * P; ++i) pi[i] = (i / P) * (P + 1) + 1 + i % P;      // drop the CLS row ('patch' select)
    RC(dalloc(&d_latidx, li.size()));
    RC(dalloc(&d_patchidx, pi.size()));
    HIPCHK(hipMemcpy(d_latidx, li.data(), li.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_patchidx, pi.data(), pi.size() * 4, hipMemcpyHostToDevice));
  }
  staged.clear();
  finalized = true;
  return 0;
}

// encode_images / project_features (llava_search_arch.py:84-94)
int vstar_vqa_engine::encode(int n, const uint16_t* pix, int first_slot) {
  if (!finalized) { set_error("weights not finalized"); return VSTAR_ERR_STATE; }
  const vstar_vqa_config& c = cfg;
  if (n <= 0 || !pix || first_slot < 0 || first_slot + n > c.max_images) { set_error("bad image slot range"); return VSTAR_ERR_INVALID; }
  HIPCHK(hipSetDevice(device));
  const int C = c.clip_hidden, H = c.llm_hidden, P = clip.P, L = c.pcv_latents, FR = P + L;
  const int inner = c.pcv_heads * c.pcv_dim_head;
  const size_t per_img = (size_t)3 * c.clip_image_size * c.clip_image_size;
  for (int i0 = 0; i0 < n; i0 += enc_batch) {
    const int nb = (n - i0 < enc_batch) ? n - i0 : enc_batch;
    const int slot0 = first_slot + i0;
    HIPCHK(hipMemcpyAsync(d_pix, pix + (size_t)i0 * per_img, (size_t)nb * per_img * 2, hipMemcpyHostToDevice, stream));
    RC(run_tower(clip, d_pix, nb));                       // clip.x = hidden_states[select_layer], [nb, P+1, C]
    // ---- long features: mm_projector on the patch rows, written straight into the feature table ----
    {
      GemmParams p{};
      p.A = clip.x; p.lda = C; p.a_group = P; p.a_gstride = P + 1; p.a_off = 1;
      p.W = proj0.W; p.bias = proj0.b; p.M = nb * P; p.N = proj0.N; p.K = proj0.K;
      if (c.projector_type == 0) {
        p.C = feats; p.ldc = H; p.c_group = P; p.c_gstride = FR; p.c_off = (int64_t)slot0 * FR;
        RC(gemm(p, VSTAR_EPI_NONE, false));
      } else {
        p.C = p_tmp; p.ldc = H;
        RC(gemm(p, VSTAR_EPI_GELU, false));
        GemmParams q{};
        q.A = p_tmp; q.lda = H; q.W = proj1.W; q.bias = proj1.b; q.M = nb * P; q.N = proj1.N; q.K = proj1.K;
        q.C = feats; q.ldc = H; q.c_group = P; q.c_gstride = FR; q.c_off = (int64_t)slot0 * FR;
        RC(gemm(q, VSTAR_EPI_NONE, false));
      }
    }
    // ---- short features: LayerNorm -> PerceiverResampler -> Linear (perceiver.py:100-121) ----
    KCHK(layernorm_lp(clip.x, pcv_ln_g, pcv_ln_b, p_nm, nb * P, C, 1e-5f, d_patchidx, 0, stream));
    KCHK(add_bcast(p_nm, pcv_media_pos, p_xm, (int64_t)nb * P, C, 1, stream));                 // x + media_pos_emb[:1]
    KCHK(gather_rows(pcv_latents, d_latidx, p_lat, nb * L, C, stream));                        // repeat(latents)
    for (int li = 0; li < c.pcv_depth; ++li) {
      PcvLayer& l = pcv[li];
      KCHK(layernorm_lp(p_xm, l.nm_g, l.nm_b, p_nm, nb * P, C, 1e-5f, nullptr, 0, stream));
      KCHK(layernorm_lp(p_lat, l.nl_g, l.nl_b, p_nl, nb * L, C, 1e-5f, nullptr, 0, stream));
      RC(lin(p_nl, C, l.to_q, p_q, inner, nb * L));
      {  // to_kv(cat(x, latents)): two GEMMs writing interleaved row groups of the [nb, P+L, 2*inner] buffer
        GemmParams p{};
        p.A = p_nm; p.lda = C; p.W = l.to_kv.W; p.M = nb * P; p.N = l.to_kv.N; p.K = l.to_kv.K;
        p.C = p_kv; p.ldc = 2 * inner; p.c_group = P; p.c_gstride = FR; p.c_off = 0;
        RC(gemm(p, VSTAR_EPI_NONE, false));
        p.A = p_nl; p.M = nb * L; p.c_group = L; p.c_off = P;
        RC(gemm(p, VSTAR_EPI_NONE, false));
      }
      KCHK(perceiver_attention(p_q, p_kv, p_att, nb, L, FR, c.pcv_heads, c.pcv_dim_head, stream));
      RC(lin(p_att, inner, l.to_out, p_lat, C, nb * L, VSTAR_EPI_NONE, p_lat, C));                 // attn(x, latents) + latents
      KCHK(layernorm_lp(p_lat, l.ff_g, l.ff_b, p_nl, nb * L, C, 1e-5f, nullptr, 0, stream));
      RC(lin(p_nl, C, l.ff1, p_ff, C * c.pcv_ff_mult, nb * L, VSTAR_EPI_GELU));
      RC(lin(p_ff, C * c.pcv_ff_mult, l.ff2, p_lat, C, nb * L, VSTAR_EPI_NONE, p_lat, C));          // ff(latents) + latents
    }
    KCHK(layernorm_lp(p_lat, pcv_norm_g, pcv_norm_b, p_nl, nb * L, C, 1e-5f, nullptr, 0, stream));
    {
      GemmParams p{};
      p.A = p_nl; p.lda = C; p.W = pcv_out.W; p.bias = pcv_out.b; p.M = nb * L; p.N = pcv_out.N; p.K = pcv_out.K;
      p.C = feats; p.ldc = H; p.c_group = L; p.c_gstride = FR; p.c_off = (int64_t)slot0 * FR + P;
      RC(gemm(p, VSTAR_EPI_NONE, false));
    }
    HIPCHK(hipStreamSynchronize(stream));    // d_pix is reused by the next chunk
  }
  return 0;
}

// =============================================== C ABI ===============================================
extern "C" {

int vstar_vqa_create(const vstar_vqa_config* cfg, int device, vstar_vqa_handle** out) {
  if (!cfg || !out) { tls_error() = "null argument"; return VSTAR_ERR_INVALID; }
  if (cfg->abi_version != VSTAR_VQA_ABI_VERSION) { tls_error() = "ABI version mismatch"; return VSTAR_ERR_INVALID; }
  if (cfg->max_slots <= 0 || cfg->max_ctx < 64 || cfg->max_rows < 64 || cfg->max_images <= 0 || cfg->pcv_latents <= 0 ||
      cfg->pcv_latents > 64 || cfg->max_ctx > 8192) {
    tls_error() = "bad limits";
    return VSTAR_ERR_INVALID;
  }
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || device < 0 || device >= n) {
    tls_error() = "no such HIP device (libvstar_hip has no CPU fallback)";
    return VSTAR_ERR_HIP;
  }
  vstar_vqa_engine* h = new vstar_vqa_engine();
  h->cfg = *cfg;
  h->device = device;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&h->stream) != hipSuccess) {
    tls_error() = "hipStreamCreate failed";
    delete h;
    return VSTAR_ERR_HIP;
  }
  *out = h;
  return VSTAR_OK;
}

void vstar_vqa_destroy(vstar_vqa_handle* h) {
  if (!h) return;
  hipSetDevice(h->device);
  hipStreamSynchronize(h->stream);
  h->release_base();
  h->run.release();
  hipStreamDestroy(h->stream);
  delete h;
}

const char* vstar_vqa_last_error(const vstar_vqa_handle* h) { return h ? h->error.c_str() : tls_error().c_str(); }

int vstar_vqa_load_tensor(vstar_vqa_handle* h, const char* key, const void* host_ptr, int dtype, int ndim, const int64_t* shape) {
  if (!h || !key || !host_ptr || ndim < 0 || ndim > 8 || (ndim && !shape)) { tls_error() = "bad argument"; return VSTAR_ERR_INVALID; }
  if (h->finalized) { h->set_error("weights already finalized"); return VSTAR_ERR_STATE; }
  return h->stage_tensor(key, host_ptr, dtype, ndim, shape);
}

int vstar_vqa_finalize_weights(vstar_vqa_handle* h) {
  if (!h) { tls_error() = "null handle"; return VSTAR_ERR_INVALID; }
  return h->finalize();
}

int vstar_vqa_encode_images(vstar_vqa_handle* h, int n, const uint16_t* pixels_f16, int first_slot) {
  if (!h) { tls_error() = "null handle"; return VSTAR_ERR_INVALID; }
  return h->encode(n, pixels_f16, first_slot);
}

int vstar_vqa_forward(vstar_vqa_handle* h, int nseq, const int32_t* row_off, const int32_t* src, const int32_t* kv_slot,
                      const int32_t* prefix_slot, const int32_t* past_len, int n_want, const int32_t* want,
                      uint16_t* logits_f16, int32_t* argmax) {
  if (!h) { tls_error() = "null handle"; return VSTAR_ERR_INVALID; }
  if (!h->finalized) { h->set_error("weights not finalized"); return VSTAR_ERR_STATE; }
  return h->run.forward(nseq, row_off, src, kv_slot, prefix_slot, past_len, n_want, want, logits_f16, argmax);
}

int vstar_vqa_op_gemm(const void* A, const void* W, const void* bias, const void* res, void* C, int M, int N, int K,
                      int epilogue, int kernel, const void* norm_w, float norm_eps) {
  if (!A || !W || !C || M <= 0 || N <= 0 || K <= 0 || K % 64) { tls_error() = "bad argument"; return VSTAR_ERR_INVALID; }
  GemmParams p{};
  const int n_out = epilogue == VSTAR_EPI_SILU_MUL ? N / 2 : N;
  p.A = (const lp_t*)A; p.lda = K; p.W = (const lp_t*)W; p.bias = (const lp_t*)bias; p.res = (const lp_t*)res; p.ldr = n_out;
  p.C = C; p.ldc = n_out; p.M = M; p.N = N; p.K = K;
  p.norm_w = (const lp_t*)norm_w; p.norm_eps = norm_eps;
  hipError_t e;
  if (kernel == 3) p.tile_force = -1;      // the register-streaming skinny kernel even where the LDS-ring variant would run
  if (kernel == 4) p.tile_force = GEMM_TILE_4W;      // round 6: the 4-wave / AGPR 256^2 kernel, error outside its domain
  if (kernel == 5) p.tile_force = 256;               // the 8-wave 256^2 kernel
  if (kernel == 1 || kernel == 3 || (kernel == 0 && gemm_skinny_eligible(p))) e = gemm_skinny_lp(p, epilogue, false, nullptr);
  else e = gemm_lp(p, epilogue, false, nullptr);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) { tls_error() = std::string("vstar_vqa_op_gemm: ") + hipGetErrorString(e); return VSTAR_ERR_HIP; }
  return VSTAR_OK;
}

int64_t vstar_vqa_debug_read(vstar_vqa_handle* h, const char* name, float* out, int64_t cap) {
  if (!h || !name || !out) { tls_error() = "bad argument"; return VSTAR_ERR_INVALID; }
  if (!h->finalized) { h->set_error("weights not finalized"); return VSTAR_ERR_STATE; }
  hipSetDevice(h->device);
  const std::string n(name);
  const lp_t* src = nullptr;
  int64_t cnt = 0;
  if (n == "features") { src = h->feats; cnt = (int64_t)h->cfg.max_images * (h->clip.P + h->cfg.pcv_latents) * h->cfg.llm_hidden; }
  else if (n == "clip_hidden") { src = h->clip.x; cnt = (int64_t)h->enc_batch * h->clip.N * h->clip.hidden; }
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

double vstar_vqa_last_forward_ms(const vstar_vqa_handle* h) { return h ? h->run.last_ms : 0.0; }

}  // extern "C"
