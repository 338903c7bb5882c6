// norm.hip — LayerNorm / RMSNorm, one wave64 per row, 16-byte lpx8 loads, row kept in registers
// (single HBM read, two-pass variance in fp32 like torch).  HBM-bound kernels.
//   LayerNorm : HF CLIPEncoderLayer.layer_norm1/2, pre_layrnorm, OWL-ViT post_layernorm/layer_norm
//               (clip_encoder.py:53-57 -> transformers CLIPVisionModel; owlvit.py:128-138), SAM nn.LayerNorm
//               (transformer.py:109-205) and LayerNorm2d over channels-last rows (common.py:31-43, eps 1e-6).
//   RMSNorm   : HF LlamaRMSNorm (fp32 variance, cast to bf16, then weight multiply) via llava_llama.py:93-102.
#include <cstdlib>
#include "common.hpp"
#include "kernels.hpp"

namespace VS_NS {

namespace {

constexpr int MAXCH = 8;  // up to 8 chunks of 64 lanes x 8 elements = 4096 columns

template <bool RMS>
__global__ __launch_bounds__(256) void norm_kernel(const lp_t* __restrict__ x, const lp_t* __restrict__ gamma,
                                                   const lp_t* __restrict__ beta, lp_t* __restrict__ y, int rows,
                                                   int cols, float eps, const int32_t* __restrict__ row_index, int act) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int64_t src = row_index ? (int64_t)row_index[row] : (int64_t)row;
  const lp_t* xr = x + src * cols;
  const int nvec = cols >> 3;
  float v[MAXCH][8];
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < MAXCH; ++c) {
    const int vi = c * 64 + lane;
    if (vi < nvec) {
      const lpx8 t = *(const lpx8*)(xr + vi * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[c][e] = lp2f((lp_t)t[e]);
        sum += RMS ? v[c][e] * v[c][e] : v[c][e];
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[c][e] = 0.f;
    }
  }
  sum = wave_sum(sum);
  float mean = 0.f, rstd;
  if (RMS) {
    rstd = rsqrtf(sum / (float)cols + eps);
  } else {
    mean = sum / (float)cols;
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < MAXCH; ++c) {
      const int vi = c * 64 + lane;
      if (vi < nvec) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = v[c][e] - mean;
          sq += d * d;
        }
      }
    }
    sq = wave_sum(sq);
    rstd = rsqrtf(sq / (float)cols + eps);
  }
  lp_t* yr = y + (int64_t)row * cols;
#pragma unroll
  for (int c = 0; c < MAXCH; ++c) {
    const int vi = c * 64 + lane;
    if (vi < nvec) {
      const lpx8 g = *(const lpx8*)(gamma + vi * 8);
      lpx8 b;
      if (!RMS && beta) b = *(const lpx8*)(beta + vi * 8);
      lpx8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float r;
        if (RMS) {
          r = lp2f((lp_t)g[e]) * rlp(v[c][e] * rstd);   // LlamaRMSNorm: weight * hidden.to(bf16)
        } else {
          r = (v[c][e] - mean) * rstd * lp2f((lp_t)g[e]) + (beta ? lp2f((lp_t)b[e]) : 0.f);
          if (act == 1) r = act_gelu_erf(rlp(r));
        }
        o[e] = (short)f2lp(r);
      }
      *(lpx8*)(yr + vi * 8) = o;
    }
  }
}

}  // namespace

hipError_t layernorm_lp(const lp_t* x, const lp_t* gamma, const lp_t* beta, lp_t* y, int rows, int cols,
                          float eps, const int32_t* row_index, int act, hipStream_t s) {
  if (rows <= 0) return hipSuccess;
  if (cols % 8 != 0 || cols > MAXCH * 512) return hipErrorInvalidValue;
  hipLaunchKernelGGL(norm_kernel<false>, dim3((rows + 3) / 4), dim3(256), 0, s, x, gamma, beta, y, rows, cols, eps,
                     row_index, act);
  return hipGetLastError();
}

namespace {
// canonical sum of squares of a 64-column span held as 8 values per lane by 8 consecutive lanes: (4 + 4 columns) per lane,
// then lane pairs (xor 1), fragments pairwise (xor 2, xor 4).  gemm256 / gemm128 epilogues use the same tree.
__device__ __forceinline__ float sumsq_span64(const float (&v)[8]) {
  const float h0 = ((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]) + v[3] * v[3];
  const float h1 = ((v[4] * v[4] + v[5] * v[5]) + v[6] * v[6]) + v[7] * v[7];
  return dpp_add_half_mirror(dpp_add_xor2(dpp_add_xor1(h0 + h1)));
}

// Across the spans of a row: 64 spans at a time, one per lane, added by a butterfly (xor 1 .. 32); passes of 64 spans are added in
// order.  rms_rstd_rows and rms_rstd_partials share this, so the two agree bit for bit.
__device__ __forceinline__ float sum_spans64(float v) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ __launch_bounds__(256) void rms_rstd_rows_kernel(const lp_t* __restrict__ x, int rows, int cols, float eps, float* __restrict__ r) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const lp_t* xr = x + (int64_t)row * cols;
  const int nspan = cols >> 6;
  float total = 0.f;
  for (int p0 = 0; p0 < nspan; p0 += 64) {             // 64 spans per pass
    float mine = 0.f;                                  // the partial sum of span p0 + lane
    for (int s0 = 0; s0 < 64; s0 += 8) {               // 8 spans per load: lane -> span p0 + s0 + lane/8, chunk lane%8
      const int sp = p0 + s0 + (lane >> 3);
      float v[8];
      if (sp < nspan) {
        const lpx8 t = *(const lpx8*)(xr + sp * 64 + (lane & 7) * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = lp2f((lp_t)t[e]);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
      }
      const float part = sumsq_span64(v);              // valid in all 8 lanes of the span
      const float got = __shfl(part, (lane & 7) * 8, 64);      // lane l (< 8 relevant) fetches span s0 + (l & 7)
      if ((lane >> 3) == (s0 >> 3)) mine = got;        // lanes s0 .. s0+7 keep spans p0 + s0 .. p0 + s0 + 7
    }
    total += sum_spans64(mine);
  }
  if (lane == 0) r[row] = rsqrtf(total / (float)cols + eps);
}

__global__ __launch_bounds__(256) void rms_rstd_partials_kernel(const float* __restrict__ partials, int ld, int rows, int cols, float eps,
                                                                float* __restrict__ r) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* pr = partials + (int64_t)row * ld;
  const int nspan = cols >> 6;
  float total = 0.f;
  for (int p0 = 0; p0 < nspan; p0 += 64) {
    const float mine = (p0 + lane < nspan) ? pr[p0 + lane] : 0.f;      // one coalesced 256-byte read per row and pass
    total += sum_spans64(mine);
  }
  if (lane == 0) r[row] = rsqrtf(total / (float)cols + eps);
}

// ---- LayerNorm folded into the consuming linear (ViT towers, round 4): variance statistics + load-time weight folding ----
__device__ __forceinline__ float sum_span64(const float (&v)[8]) {
  const float h0 = ((v[0] + v[1]) + v[2]) + v[3];
  const float h1 = ((v[4] + v[5]) + v[6]) + v[7];
  return dpp_add_half_mirror(dpp_add_xor2(dpp_add_xor1(h0 + h1)));
}

__device__ __forceinline__ float ln_rstd_from(float total_sq, float total_s, int cols, float eps) {
  const float mean = total_s / (float)cols;
  const float var = fmaxf(total_sq / (float)cols - mean * mean, 0.f);
  return rsqrtf(var + eps);
}

__global__ __launch_bounds__(256) void ln_rstd_rows_kernel(const lp_t* __restrict__ x, int rows, int cols, float eps, float* __restrict__ r) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const lp_t* xr = x + (int64_t)row * cols;
  const int nspan = cols >> 6;
  float total_sq = 0.f, total_s = 0.f;
  for (int p0 = 0; p0 < nspan; p0 += 64) {
    float mine_sq = 0.f, mine_s = 0.f;
    for (int s0 = 0; s0 < 64; s0 += 8) {
      const int sp = p0 + s0 + (lane >> 3);
      float v[8];
      if (sp < nspan) {
        const lpx8 t = *(const lpx8*)(xr + sp * 64 + (lane & 7) * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = lp2f((lp_t)t[e]);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
      }
      const float psq = sumsq_span64(v), ps = sum_span64(v);
      const float gsq = __shfl(psq, (lane & 7) * 8, 64), gs = __shfl(ps, (lane & 7) * 8, 64);
      if ((lane >> 3) == (s0 >> 3)) { mine_sq = gsq; mine_s = gs; }
    }
    total_sq += sum_spans64(mine_sq);
    total_s += sum_spans64(mine_s);
  }
  if (lane == 0) r[row] = ln_rstd_from(total_sq, total_s, cols, eps);
}

__global__ __launch_bounds__(256) void ln_rstd_partials_kernel(const float* __restrict__ partials, int ld, int rows, int cols, float eps,
                                                               float* __restrict__ r) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* pr = partials + (int64_t)row * ld;
  const int nspan = cols >> 6;
  float total_sq = 0.f, total_s = 0.f;
  for (int p0 = 0; p0 < nspan; p0 += 64) {
    const bool on = p0 + lane < nspan;
    total_sq += sum_spans64(on ? pr[p0 + lane] : 0.f);
    total_s += sum_spans64(on ? pr[nspan + p0 + lane] : 0.f);
  }
  if (lane == 0) r[row] = ln_rstd_from(total_sq, total_s, cols, eps);
}

// one workgroup per packed weight row: bias += W . b_ln (unscaled W, fp32), then W := W * g - mean_k(W * g), rounded once.
// ZERO-SUM ROUNDING (round 5): the consumer drops the row mean of x because sum_k W'_k = 0 — true of the centred fp32
// row, not of its 16-bit rounding: the leftover c = sum_k round(W'_k) (~ sqrt(K) / 2 ulp) re-enters every output as
// rstd * mean(x) * c, an error that grows with |mean| / std of the activation row (massive-activation tokens of real CLIP / OWL
// checkpoints).  So after the nearest rounding, thread 0 moves a few elements (~15 of 768) to their OTHER 16-bit neighbour —
// those whose rounding error points the way the residual does, nearest-to-the-tie first, never past the residual — until the
// ROUNDED row sums to zero within the smallest step used (measured: |c| 1.9e-3 -> 2.7e-7 on N(0, 0.03) rows, +0.7 % rms weight
// rounding noise).  No run-time term, no extra epilogue operand; both kernels read the same weights, so bit-identity between
// tile shapes is untouched.
// (one WAVE per row: the zero-sum pass is serial in one lane, so small workgroups keep more rows in flight per CU — load-time only)
__global__ __launch_bounds__(64) void ln_fold_kernel(lp_t* __restrict__ W, lp_t* __restrict__ bias, const lp_t* __restrict__ g,
                                                      const lp_t* __restrict__ b_ln, int K, int zero_sum) {
  extern __shared__ __attribute__((aligned(16))) char fold_smem[];
  float* ex = (float*)fold_smem;                       // [K] the centred fp32 row
  lp_t* qb = (lp_t*)(fold_smem + (size_t)K * 4);       // [K] its 16-bit rounding
  const int n = blockIdx.x, tid = threadIdx.x;
  lp_t* w = W + (int64_t)n * K;
  float dot = 0.f, sum = 0.f;
  // (four interleaved partial sums per lane, combined pairwise, then across the wave: the summation order of the 256-thread form)
  float dp[4] = {0.f, 0.f, 0.f, 0.f}, sp[4] = {0.f, 0.f, 0.f, 0.f};
  for (int k = tid, q = 0; k < K; k += 64, q = (q + 1) & 3) {
    const float wv = lp2f(w[k]);
    dp[q] += wv * lp2f(b_ln[k]);
    sp[q] += wv * lp2f(g[k]);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) { dp[q] = wave_sum(dp[q]); sp[q] = wave_sum(sp[q]); }
  dot = (dp[0] + dp[1]) + (dp[2] + dp[3]);
  sum = (sp[0] + sp[1]) + (sp[2] + sp[3]);
  const float mean = sum / (float)K;
  for (int k = tid; k < K; k += 64) {
    const float e = lp2f(w[k]) * lp2f(g[k]) - mean;
    ex[k] = e;
    qb[k] = f2lp(e);
  }
  __syncthreads();
  if (zero_sum && tid == 0) {
    double r = 0.0;
    for (int k = 0; k < K; ++k) r += (double)lp2f(qb[k]);
    const float thr[5] = {0.4f, 0.3f, 0.2f, 0.1f, 0.0f};
    for (int t = 0; t < 5 && r != 0.0; ++t) {
      for (int k = 0; k < K && r != 0.0; ++k) {
        const lp_t q = qb[k];
        if (q & 1u << 15 ? (q & 0x7fff) == 0 : q == 0) continue;       // +-0: no neighbour on one side in this bit walk
        const float qv = lp2f(q), err = qv - ex[k];
        if (err == 0.f || (err > 0.f) != (r > 0.0)) continue;          // only elements rounded the way the residual points
        const bool down = r > 0.0;                                     // move this element down (r > 0) or up
        const bool neg = (q >> 15) != 0;
        const lp_t nb = (lp_t)((down != neg) ? q - 1 : q + 1);          // next representable value in that direction
        const float nv = lp2f(nb);
        if (!(nv == nv) || fabsf(nv) > 3.0e38f) continue;
        const double d = (double)nv - (double)qv;
        // was already moved once?  (|new error| would exceed one step): each element moves at most once
        if (fabsf(err) > fabsf((float)d) * 0.5001f) continue;
        if (fabsf(err) < thr[t] * fabsf((float)d)) continue;
        if (fabs(r + d) < fabs(r)) { qb[k] = nb; r += d; }
      }
    }
  }
  __syncthreads();
  for (int k = tid; k < K; k += 64) w[k] = qb[k];
  if (tid == 0 && bias) bias[n] = f2lp(lp2f(bias[n]) + dot);
}

__global__ void scale_cols_kernel(lp_t* __restrict__ W, const lp_t* __restrict__ w, int64_t n_vec, int kv) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_vec) return;
  const int c = (int)(idx % kv);
  lpx8 a = *(const lpx8*)(W + idx * 8);
  const lpx8 g = *(const lpx8*)(w + (int64_t)c * 8);
#pragma unroll
  for (int e = 0; e < 8; ++e) a[e] = (short)f2lp(lp2f((lp_t)a[e]) * lp2f((lp_t)g[e]));
  *(lpx8*)(W + idx * 8) = a;
}

__global__ void fill_lp_kernel(lp_t* __restrict__ v, int64_t n, lp_t value) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < n) v[idx] = value;
}
}  // namespace

hipError_t rms_rstd_rows(const lp_t* x, int rows, int cols, float eps, float* r, hipStream_t s) {
  if (cols % 64 || rows <= 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(rms_rstd_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, rows, cols, eps, r);
  return hipGetLastError();
}

hipError_t rms_rstd_partials(const float* partials, int ld, int rows, int cols, float eps, float* r, hipStream_t s) {
  if (cols % 64 || rows <= 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(rms_rstd_partials_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, partials, ld, rows, cols, eps, r);
  return hipGetLastError();
}

hipError_t ln_rstd_rows(const lp_t* x, int rows, int cols, float eps, float* r, hipStream_t s) {
  if (cols % 64 || rows <= 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(ln_rstd_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, rows, cols, eps, r);
  return hipGetLastError();
}

hipError_t ln_rstd_partials(const float* partials, int ld, int rows, int cols, float eps, float* r, hipStream_t s) {
  if (cols % 64 || rows <= 0 || ld < 2 * (cols / 64)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(ln_rstd_partials_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, partials, ld, rows, cols, eps, r);
  return hipGetLastError();
}

hipError_t ln_fold_weights(lp_t* W, lp_t* bias, const lp_t* g, const lp_t* b_ln, int n_rows, int K, hipStream_t s) {
  if (n_rows <= 0 || K <= 0 || K > 8192) return hipErrorInvalidValue;
  // VSTAR_FOLD_ZERO_SUM=0 (A/B runs): plain nearest rounding of the centred rows
  static const int zero_sum = [] { const char* e = getenv("VSTAR_FOLD_ZERO_SUM"); return (e && atoi(e) == 0) ? 0 : 1; }();
  hipLaunchKernelGGL(ln_fold_kernel, dim3(n_rows), dim3(64), (size_t)K * 6, s, W, bias, g, b_ln, K, zero_sum);
  return hipGetLastError();
}

hipError_t scale_cols_lp(lp_t* W, const lp_t* w, int64_t rows, int K, hipStream_t s) {
  if (K % 8 || rows <= 0) return hipErrorInvalidValue;
  const int64_t n_vec = rows * (K / 8);
  hipLaunchKernelGGL(scale_cols_kernel, dim3((unsigned)((n_vec + 255) / 256)), dim3(256), 0, s, W, w, n_vec, K / 8);
  return hipGetLastError();
}

hipError_t fill_lp(lp_t* v, int64_t n, float value, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(fill_lp_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, v, n, f2lp(value));
  return hipGetLastError();
}

hipError_t rmsnorm_lp(const lp_t* x, const lp_t* gamma, lp_t* y, int rows, int cols, float eps,
                        const int32_t* row_index, hipStream_t s) {
  if (rows <= 0) return hipSuccess;
  if (cols % 8 != 0 || cols > MAXCH * 512) return hipErrorInvalidValue;
  hipLaunchKernelGGL(norm_kernel<true>, dim3((rows + 3) / 4), dim3(256), 0, s, x, gamma, (const lp_t*)nullptr, y, rows,
                     cols, eps, row_index, 0);
  return hipGetLastError();
}

}  // namespace VS_NS
