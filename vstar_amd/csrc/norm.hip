// norm.hip — LayerNorm / RMSNorm, one wave64 per row, 16-byte lpx8 loads, row kept in registers
// (single HBM read, two-pass variance in fp32 like torch).  HBM-bound kernels.
//   LayerNorm : HF CLIPEncoderLayer.layer_norm1/2, pre_layrnorm, OWL-ViT post_layernorm/layer_norm
//               (clip_encoder.py:53-57 -> transformers CLIPVisionModel; owlvit.py:128-138), SAM nn.LayerNorm
//               (transformer.py:109-205) and LayerNorm2d over channels-last rows (common.py:31-43, eps 1e-6).
//   RMSNorm   : HF LlamaRMSNorm (fp32 variance, cast to bf16, then weight multiply) via llava_llama.py:93-102.
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

hipError_t rmsnorm_lp(const lp_t* x, const lp_t* gamma, lp_t* y, int rows, int cols, float eps,
                        const int32_t* row_index, hipStream_t s) {
  if (rows <= 0) return hipSuccess;
  if (cols % 8 != 0 || cols > MAXCH * 512) return hipErrorInvalidValue;
  hipLaunchKernelGGL(norm_kernel<true>, dim3((rows + 3) / 4), dim3(256), 0, s, x, gamma, (const lp_t*)nullptr, y, rows,
                     cols, eps, row_index, 0);
  return hipGetLastError();
}

}  // namespace VS_NS
