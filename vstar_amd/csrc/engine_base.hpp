// engine_base.hpp — what the two engines of libvstar_hip.so share: checkpoint staging, weight packing for the GEMM kernels,
// GEMM launches with HIP-event profiling, and the HF CLIP/OWL-ViT vision tower (weights + forward).  Like the kernel
// files it is dtype-generic over lp_t and lives in VS_NS: engine.hip (VSM, bf16) and vqa_engine.hip (VQA-LLM, built with
// -DVSTAR_LP_F16) each get their own instantiation.
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>
#include "common.hpp"
#include "kernels.hpp"

namespace VS_NS {

inline std::string& tls_error() {
  static thread_local std::string s;
  return s;
}

struct HostTensor {
  std::vector<lp_t> data;
  std::vector<int64_t> shape;
  int64_t numel() const { int64_t n = 1; for (auto d : shape) n *= d; return n; }
};

struct Lin { lp_t* W = nullptr; lp_t* b = nullptr; int N = 0, K = 0; };
struct VitBlock { lp_t *ln1_g, *ln1_b, *ln2_g, *ln2_b; Lin qkv, out, fc1, fc2; };
struct VitTower {
  int image = 0, patch = 0, grid = 0, P = 0, N = 0, hidden = 0, heads = 0, mlp = 0, nblocks = 0, kpad = 0;
  Lin patch_lin; lp_t *cls = nullptr, *pos = nullptr, *pre_g = nullptr, *pre_b = nullptr;
  std::vector<VitBlock> blocks;
  // activations
  lp_t *im2col = nullptr, *patch_out = nullptr, *x = nullptr, *h = nullptr, *qkv = nullptr, *att = nullptr, *mlp_buf = nullptr;
  // LayerNorms folded into the q|k|v and fc1 linears (round 4; build_tower): per-row statistics of the residual stream — the
  // epilogues that write it emit 64-column sums of squares and sums — and 1 / sqrt(var + eps) per row; no normalised copy of x
  bool fold = false;
  float *part = nullptr, *rstd = nullptr;
};
// fp8 twin of a packed Linear (W8A8 mode): e4m3 rows + per-output-channel scales, same row order/padding as Lin::W
struct Lin8 { uint8_t* W = nullptr; float* s = nullptr; };
struct LlmBlock {
  lp_t *in_norm, *post_norm; Lin qkv, o, gate_up, down; Lin8 qkv8, o8, gate_up8, down8;
  Lin8 qkv8f, gate_up8f;      // W8A8, fully block-scaled chain: fp8 of W diag(norm weight) (the RMSNorm folded into the linear)
};

#define HIPCHK(expr)                                                                         \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess) {                                                                  \
      set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                          \
      return VSTAR_ERR_HIP;                                                                  \
    }                                                                                        \
  } while (0)
#define KCHK(expr) HIPCHK(expr)
#define RC(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)

struct EngineBase {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string error;
  bool finalized = false;
  std::map<std::string, HostTensor> staged;
  std::vector<void*> allocs;

  // profiling
  bool profile = false;
  std::vector<hipEvent_t> ev;
  size_t ev_used = 0;
  double prof_ms = 0, prof_flops = 0;
  int64_t prof_launches = 0;
  double pending_flops = 0;
  // the W8A8 launches (fp8 MFMA) of the same profile, kept apart: their peak is twice the bf16 one
  double prof_ms8 = 0, prof_flops8 = 0;
  int64_t prof_launches8 = 0;
  std::vector<double> ev_flops;        // per recorded launch: its FLOPs, negative when it ran on the fp8 MFMA

  void set_error(const std::string& m) { error = m; tls_error() = m; }

  // Converts one checkpoint tensor (fp32 / fp16 / bf16 host data) to the engine's storage type and stages it by key.
  int stage_tensor(const char* key, const void* host_ptr, int dtype, int ndim, const int64_t* shape);
  void release_base() {
    for (void* p : allocs) hipFree(p);
    allocs.clear();
    for (auto e : ev) hipEventDestroy(e);
    ev.clear();
  }

  template <typename T> int dalloc(T** p, size_t count) {
    void* q = nullptr;
    size_t bytes = count * sizeof(T);
    if (bytes == 0) bytes = 16;
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) { set_error("hipMalloc(" + std::to_string(bytes) + "): " + hipGetErrorString(e)); return VSTAR_ERR_NOMEM; }
    allocs.push_back(q);
    *p = (T*)q;
    return 0;
  }
  const HostTensor* find(const std::string& k) {
    auto it = staged.find(k);
    return it == staged.end() ? nullptr : &it->second;
  }
  int need(const std::string& k, const HostTensor** out) {
    *out = find(k);
    if (!*out) { set_error("missing checkpoint tensor: " + k); return VSTAR_ERR_MISSING; }
    return 0;
  }
  int upload_vec(const std::string& k, lp_t** dev, int64_t expect = -1) {
    const HostTensor* t;
    int rc = need(k, &t);
    if (rc) return rc;
    if (expect >= 0 && t->numel() != expect) { set_error("bad size for " + k); return VSTAR_ERR_INVALID; }
    rc = dalloc(dev, (size_t)t->numel());
    if (rc) return rc;
    HIPCHK(hipMemcpy(*dev, t->data.data(), t->numel() * 2, hipMemcpyHostToDevice));
    return 0;
  }
  // Packs rows of several [n_i, K] matrices (optionally with biases) into one padded [Npad, Kpad] device matrix.
  // perm: optional row permutation of the concatenated matrix (packed row r <- concat row perm[r]).
  int make_lin(const std::vector<std::string>& wkeys, const std::vector<std::string>& bkeys, Lin* out, int K_expect = -1,
               const std::vector<int>* perm = nullptr, bool conv3x3 = false, bool conv_patch = false) {
    std::vector<const HostTensor*> ws;
    int N = 0, K = -1;
    for (auto& k : wkeys) {
      const HostTensor* t;
      int rc = need(k, &t);
      if (rc) return rc;
      int n = (int)t->shape[0];
      int kk = (int)(t->numel() / n);
      if (K < 0) K = kk;
      if (kk != K) { set_error("K mismatch in " + k); return VSTAR_ERR_INVALID; }
      ws.push_back(t);
      N += n;
    }
    if (K_expect >= 0 && K != K_expect) { set_error("unexpected K for " + wkeys[0]); return VSTAR_ERR_INVALID; }
    const int Kpad = (K + 63) / 64 * 64, Npad = (N + 255) / 256 * 256;
    std::vector<lp_t> host((size_t)Npad * Kpad, 0);
    int r0 = 0;
    for (auto* t : ws) {
      const int n = (int)t->shape[0];
      for (int r = 0; r < n; ++r) {
        lp_t* dst = &host[(size_t)(r0 + r) * Kpad];
        const lp_t* src = &t->data[(size_t)r * K];
        if (conv3x3) {   // [O][C][3][3] -> [O][(ky*3+kx)*C + c]
          const int C = K / 9;
          for (int c = 0; c < C; ++c)
            for (int tap = 0; tap < 9; ++tap) dst[tap * C + c] = src[c * 9 + tap];
        } else {
          (void)conv_patch;  // [O][C][ky][kx] flattens to c*ps*ps + ky*ps + kx, which is what im2col_patch emits
          memcpy(dst, src, (size_t)K * 2);
        }
      }
      r0 += n;
    }
    if (perm) {
      std::vector<lp_t> tmp((size_t)Npad * Kpad, 0);
      for (int r = 0; r < N; ++r) memcpy(&tmp[(size_t)r * Kpad], &host[(size_t)(*perm)[r] * Kpad], (size_t)Kpad * 2);
      host.swap(tmp);
    }
    int rc = dalloc(&out->W, host.size());
    if (rc) return rc;
    HIPCHK(hipMemcpy(out->W, host.data(), host.size() * 2, hipMemcpyHostToDevice));
    out->N = N;
    out->K = Kpad;
    out->b = nullptr;
    if (!bkeys.empty()) {
      std::vector<lp_t> hb((size_t)Npad, 0);
      int o = 0;
      for (auto& k : bkeys) {
        const HostTensor* t;
        rc = need(k, &t);
        if (rc) return rc;
        memcpy(&hb[o], t->data.data(), (size_t)t->numel() * 2);
        o += (int)t->numel();
      }
      if (o != N) { set_error("bias size mismatch for " + wkeys[0]); return VSTAR_ERR_INVALID; }
      if (perm) {
        std::vector<lp_t> tb((size_t)Npad, 0);
        for (int r = 0; r < N; ++r) tb[r] = hb[(*perm)[r]];
        hb.swap(tb);
      }
      rc = dalloc(&out->b, hb.size());
      if (rc) return rc;
      HIPCHK(hipMemcpy(out->b, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
    }
    return 0;
  }

  // ---- GEMM launch with optional event profiling ----
  int gemm(const GemmParams& p, int epi, bool f32) {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (profile) {
      if (ev_used + 2 > ev.size()) {
        for (int i = 0; i < 256; ++i) { hipEvent_t e; hipEventCreate(&e); ev.push_back(e); }
      }
      e0 = ev[ev_used++]; e1 = ev[ev_used++];
      hipEventRecord(e0, stream);
    }
    hipError_t e = gemm_lp(p, epi, f32, stream);
    if (e != hipSuccess) { set_error(std::string("gemm launch: ") + hipGetErrorString(e)); return VSTAR_ERR_HIP; }
    if (profile) {
      hipEventRecord(e1, stream);
      const double fl = 2.0 * p.M * (double)p.N * p.K;
      pending_flops += fl;
      ev_flops.push_back((p.a_scale || p.a_mx) ? -fl : fl);
      prof_launches++;
      if (p.a_scale || p.a_mx) prof_launches8++;
    }
    return 0;
  }
  int lin(const lp_t* A, int64_t lda, const Lin& L, void* C, int64_t ldc, int M, int epi = VSTAR_EPI_NONE,
          const lp_t* res = nullptr, int64_t ldr = 0, bool f32 = false) {
    GemmParams p{};
    p.A = A; p.lda = lda; p.a_group = 0;
    p.W = L.W; p.bias = L.b; p.res = res; p.ldr = ldr;
    p.C = C; p.ldc = ldc; p.c_group = 0;
    p.M = M; p.N = L.N; p.K = L.K;
    p.row_scale = next_row_scale; p.sumsq_out = next_sumsq; p.sumsq_ld = next_sumsq_ld;     // one-shot (folded norms)
    p.stats_sum = next_stats_sum;
    next_row_scale = nullptr; next_sumsq = nullptr; next_sumsq_ld = 0; next_stats_sum = 0;
    if (map_group > 0) {      // row map in force (shared-prefix LLaMA pass): compact row r -> (r / group) * gstride + off + r % group
      p.a_group = p.c_group = map_group;
      p.a_gstride = p.c_gstride = map_gstride;
      p.a_off = p.c_off = map_off;
    }
    return gemm(p, epi, f32);
  }
  int map_group = 0; int64_t map_gstride = 0, map_off = 0;
  const float* next_row_scale = nullptr; float* next_sumsq = nullptr; int next_sumsq_ld = 0;   // consumed by the next lin()
  int next_stats_sum = 0;
  bool fold_vit_norms = true;          // VSTAR_FOLD_NORMS=0 / VSTAR_FOLD_VIT_NORMS=0: the LayerNorm kernels (the reference's rounding points)
  void collect_profile() {
    if (!profile) return;
    for (size_t i = 0; i + 1 < ev_used; i += 2) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, ev[i], ev[i + 1]) == hipSuccess) {
        prof_ms += ms;
        if (i / 2 < ev_flops.size() && ev_flops[i / 2] < 0) { prof_ms8 += ms; prof_flops8 += -ev_flops[i / 2]; }
      }
    }
    prof_flops += pending_flops;
    pending_flops = 0;
    ev_flops.clear();
    ev_used = 0;
  }

  int build_tower(VitTower& t, const std::string& pre, const std::string& preln_name, int image, int patch, int hidden,
                  int heads, int mlp, int nblocks, int maxB);
  int run_tower(VitTower& t, const lp_t* pix, int B);
};

inline float half_bits_to_float(uint16_t hb) {
  const uint32_t sign = (hb & 0x8000u) << 16;
  uint32_t exp = (hb >> 10) & 0x1f, man = hb & 0x3ff;
  uint32_t bits;
  if (exp == 0) {
    if (man == 0) bits = sign;
    else { int sh = 0; while (!(man & 0x400)) { man <<= 1; sh++; } man &= 0x3ff; bits = sign | ((127 - 15 - sh + 1) << 23) | (man << 13); }
  } else if (exp == 31) bits = sign | 0x7f800000u | (man << 13);
  else bits = sign | ((exp - 15 + 127) << 23) | (man << 13);
  float f; memcpy(&f, &bits, 4);
  return f;
}
inline float bf16_bits_to_float(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }

inline int EngineBase::stage_tensor(const char* key, const void* host_ptr, int dtype, int ndim, const int64_t* shape) {
  HostTensor t;
  t.shape.assign(shape, shape + ndim);
  const int64_t n = t.numel();
  t.data.resize((size_t)n);
#ifdef VSTAR_LP_F16
  const int native = VSTAR_F16, other = VSTAR_BF16;
#else
  const int native = VSTAR_BF16, other = VSTAR_F16;
#endif
  if (dtype == native) {
    memcpy(t.data.data(), host_ptr, (size_t)n * 2);
  } else if (dtype == VSTAR_F32) {
    const float* s = (const float*)host_ptr;
    for (int64_t i = 0; i < n; ++i) t.data[i] = f2lp(s[i]);
  } else if (dtype == other) {
    const uint16_t* s = (const uint16_t*)host_ptr;
    for (int64_t i = 0; i < n; ++i) t.data[i] = f2lp(other == VSTAR_F16 ? half_bits_to_float(s[i]) : bf16_bits_to_float(s[i]));
  } else { set_error("unknown dtype"); return VSTAR_ERR_INVALID; }
  staged[key] = std::move(t);
  return VSTAR_OK;
}

inline int EngineBase::build_tower(VitTower& t, const std::string& pre, const std::string& preln_name, int image, int patch,
                              int hidden, int heads, int mlp, int nblocks, int maxB) {
  t.image = image; t.patch = patch; t.grid = image / patch; t.P = t.grid * t.grid; t.N = t.P + 1;
  t.hidden = hidden; t.heads = heads; t.mlp = mlp; t.nblocks = nblocks;
  if (hidden != heads * 64) { set_error("ViT head dim must be 64"); return VSTAR_ERR_INVALID; }
  RC(make_lin({pre + "embeddings.patch_embedding.weight"}, {}, &t.patch_lin, 3 * patch * patch));
  t.kpad = t.patch_lin.K;
  RC(upload_vec(pre + "embeddings.class_embedding", &t.cls, hidden));
  RC(upload_vec(pre + "embeddings.position_embedding.weight", &t.pos, (int64_t)t.N * hidden));
  RC(upload_vec(pre + preln_name + ".weight", &t.pre_g, hidden));
  RC(upload_vec(pre + preln_name + ".bias", &t.pre_b, hidden));
  t.blocks.resize(nblocks);
  for (int i = 0; i < nblocks; ++i) {
    const std::string lp = pre + "encoder.layers." + std::to_string(i) + ".";
    VitBlock& b = t.blocks[i];
    RC(upload_vec(lp + "layer_norm1.weight", &b.ln1_g, hidden));
    RC(upload_vec(lp + "layer_norm1.bias", &b.ln1_b, hidden));
    RC(upload_vec(lp + "layer_norm2.weight", &b.ln2_g, hidden));
    RC(upload_vec(lp + "layer_norm2.bias", &b.ln2_b, hidden));
    RC(make_lin({lp + "self_attn.q_proj.weight", lp + "self_attn.k_proj.weight", lp + "self_attn.v_proj.weight"},
                {lp + "self_attn.q_proj.bias", lp + "self_attn.k_proj.bias", lp + "self_attn.v_proj.bias"}, &b.qkv, hidden));
    RC(make_lin({lp + "self_attn.out_proj.weight"}, {lp + "self_attn.out_proj.bias"}, &b.out, hidden));
    RC(make_lin({lp + "mlp.fc1.weight"}, {lp + "mlp.fc1.bias"}, &b.fc1, hidden));
    RC(make_lin({lp + "mlp.fc2.weight"}, {lp + "mlp.fc2.bias"}, &b.fc2, mlp));
  }
  // Linear(LayerNorm(x)) = rstd(x) * (x . W'^T) + (W b_ln + c), W' = W diag(g) with every row centred over k (the mean of x then
  // drops out of the product): one bf16 rounding of W' in place of the reference's rounding of the normalised activations, no
  // normalised copy of x written or read, no LayerNorm launch — the ViT twin of the RMSNorm fold of llm_forward.
  for (const char* e : {"VSTAR_FOLD_NORMS", "VSTAR_FOLD_VIT_NORMS"})
    if (const char* v = getenv(e)) if (atoi(v) == 0) fold_vit_norms = false;
  t.fold = fold_vit_norms && hidden % 64 == 0;
  if (t.fold) {
    for (auto& b : t.blocks) {
      if (b.qkv.K != hidden || b.fc1.K != hidden || !b.qkv.b || !b.fc1.b) { t.fold = false; break; }
    }
  }
  if (t.fold) {
    for (auto& b : t.blocks) {
      KCHK(ln_fold_weights(b.qkv.W, b.qkv.b, b.ln1_g, b.ln1_b, b.qkv.N, b.qkv.K, stream));
      KCHK(ln_fold_weights(b.fc1.W, b.fc1.b, b.ln2_g, b.ln2_b, b.fc1.N, b.fc1.K, stream));
    }
    HIPCHK(hipStreamSynchronize(stream));
  }
  const size_t rows = (size_t)maxB * t.N;
  RC(dalloc(&t.im2col, (size_t)maxB * t.P * t.kpad));
  RC(dalloc(&t.patch_out, (size_t)maxB * t.P * hidden));
  RC(dalloc(&t.x, rows * hidden));
  RC(dalloc(&t.h, rows * hidden));
  RC(dalloc(&t.qkv, rows * 3 * hidden));
  RC(dalloc(&t.att, rows * hidden));
  RC(dalloc(&t.mlp_buf, rows * mlp));
  if (t.fold) {
    RC(dalloc(&t.part, rows * (size_t)(2 * (hidden / 64))));
    RC(dalloc(&t.rstd, rows));
  }
  return 0;
}

// HF CLIPVisionTransformer / OwlViTVisionTransformer forward up to the last executed block (pre-LN blocks, quick-GELU)
inline int EngineBase::run_tower(VitTower& t, const lp_t* pix, int B) {
  const int C = t.hidden, rows = B * t.N;
  KCHK(im2col_patch(pix, t.im2col, B, t.image, t.patch, t.kpad, stream));
  RC(lin(t.im2col, t.kpad, t.patch_lin, t.patch_out, C, B * t.P));
  KCHK(vit_assemble_tokens(t.patch_out, t.cls, t.pos, t.h, B, t.P, C, stream));
  KCHK(layernorm_lp(t.h, t.pre_g, t.pre_b, t.x, rows, C, 1e-5f, nullptr, 0, stream));
  if (t.fold) {
    // folded LayerNorms: q|k|v and fc1 read the residual stream itself, scaled per row by 1 / sqrt(var + eps); the statistics come
    // from the rows (first block: the pre-LayerNorm's output) or from the partial sums the epilogue that wrote the stream left
    const int nsp = C / 64;
    for (int i = 0; i < t.nblocks; ++i) {
      VitBlock& b = t.blocks[i];
      if (i == 0) KCHK(ln_rstd_rows(t.x, rows, C, 1e-5f, t.rstd, stream));
      else KCHK(ln_rstd_partials(t.part, 2 * nsp, rows, C, 1e-5f, t.rstd, stream));
      next_row_scale = t.rstd;
      RC(lin(t.x, C, b.qkv, t.qkv, 3 * C, rows));
      KCHK(attn_forward(t.qkv, t.att, B, t.N, t.heads, 64, 0, 0.125f, stream));
      next_sumsq = t.part; next_sumsq_ld = 2 * nsp; next_stats_sum = nsp;
      RC(lin(t.att, C, b.out, t.x, C, rows, VSTAR_EPI_NONE, t.x, C));
      KCHK(ln_rstd_partials(t.part, 2 * nsp, rows, C, 1e-5f, t.rstd, stream));
      next_row_scale = t.rstd;
      RC(lin(t.x, C, b.fc1, t.mlp_buf, t.mlp, rows, VSTAR_EPI_QUICK_GELU));
      if (i + 1 < t.nblocks) { next_sumsq = t.part; next_sumsq_ld = 2 * nsp; next_stats_sum = nsp; }
      RC(lin(t.mlp_buf, t.mlp, b.fc2, t.x, C, rows, VSTAR_EPI_NONE, t.x, C));
    }
    return 0;
  }
  for (int i = 0; i < t.nblocks; ++i) {
    VitBlock& b = t.blocks[i];
    KCHK(layernorm_lp(t.x, b.ln1_g, b.ln1_b, t.h, rows, C, 1e-5f, nullptr, 0, stream));
    RC(lin(t.h, C, b.qkv, t.qkv, 3 * C, rows));
    KCHK(attn_forward(t.qkv, t.att, B, t.N, t.heads, 64, 0, 0.125f, stream));
    RC(lin(t.att, C, b.out, t.x, C, rows, VSTAR_EPI_NONE, t.x, C));
    KCHK(layernorm_lp(t.x, b.ln2_g, b.ln2_b, t.h, rows, C, 1e-5f, nullptr, 0, stream));
    RC(lin(t.h, C, b.fc1, t.mlp_buf, t.mlp, rows, VSTAR_EPI_QUICK_GELU));
    RC(lin(t.mlp_buf, t.mlp, b.fc2, t.x, C, rows, VSTAR_EPI_NONE, t.x, C));
  }
  return 0;
}

}  // namespace VS_NS
using namespace VS_NS;
