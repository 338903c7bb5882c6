// norm.hip — LayerNorm / RMSNorm, one wave64 per row, 16-byte bf16x8 loads, row kept in registers
// (single HBM read, two-pass variance in fp32 like torch).  HBM-bound kernels.
//   LayerNorm : HF CLIPEncoderLayer.layer_norm1/2, pre_layrnorm, OWL-ViT post_layernorm/layer_norm
//               (clip_encoder.py:53-57 -> transformers CLIPVisionModel; owlvit.py:128-138), SAM nn.LayerNorm
//               (transformer.py:109-205) and LayerNorm2d over channels-last rows (common.py:31-43, eps 1e-6).
//   RMSNorm   : HF LlamaRMSNorm (fp32 variance, cast to bf16, then weight multiply) via llava_llama.py:93-102.
#include "common.hpp"
#include "kernels.hpp"

namespace {

constexpr int MAXCH = 8;  // up to 8 chunks of 64 lanes x 8 elements = 4096 columns

template <bool RMS>
__global__ __launch_bounds__(256) void norm_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ gamma,
                                                   const bf16_t* __restrict__ beta, bf16_t* __restrict__ y, int rows,
                                                   int cols, float eps, const int32_t* __restrict__ row_index, int act) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int64_t src = row_index ? (int64_t)row_index[row] : (int64_t)row;
  const bf16_t* xr = x + src * cols;
  const int nvec = cols >> 3;
  float v[MAXCH][8];
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < MAXCH; ++c) {
    const int vi = c * 64 + lane;
    if (vi < nvec) {
      const bf16x8 t = *(const bf16x8*)(xr + vi * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[c][e] = bf2f((bf16_t)t[e]);
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
  bf16_t* yr = y + (int64_t)row * cols;
#pragma unroll
  for (int c = 0; c < MAXCH; ++c) {
    const int vi = c * 64 + lane;
    if (vi < nvec) {
      const bf16x8 g = *(const bf16x8*)(gamma + vi * 8);
      bf16x8 b;
      if (!RMS && beta) b = *(const bf16x8*)(beta + vi * 8);
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float r;
        if (RMS) {
          r = bf2f((bf16_t)g[e]) * rbf(v[c][e] * rstd);   // LlamaRMSNorm: weight * hidden.to(bf16)
        } else {
          r = (v[c][e] - mean) * rstd * bf2f((bf16_t)g[e]) + (beta ? bf2f((bf16_t)b[e]) : 0.f);
          if (act == 1) r = act_gelu_erf(rbf(r));
        }
        o[e] = (short)f2bf(r);
      }
      *(bf16x8*)(yr + vi * 8) = o;
    }
  }
}

}  // namespace

hipError_t layernorm_bf16(const bf16_t* x, const bf16_t* gamma, const bf16_t* beta, bf16_t* y, int rows, int cols,
                          float eps, const int32_t* row_index, int act, hipStream_t s) {
  if (rows <= 0) return hipSuccess;
  if (cols % 8 != 0 || cols > MAXCH * 512) return hipErrorInvalidValue;
  hipLaunchKernelGGL(norm_kernel<false>, dim3((rows + 3) / 4), dim3(256), 0, s, x, gamma, beta, y, rows, cols, eps,
                     row_index, act);
  return hipGetLastError();
}

hipError_t rmsnorm_bf16(const bf16_t* x, const bf16_t* gamma, bf16_t* y, int rows, int cols, float eps,
                        const int32_t* row_index, hipStream_t s) {
  if (rows <= 0) return hipSuccess;
  if (cols % 8 != 0 || cols > MAXCH * 512) return hipErrorInvalidValue;
  hipLaunchKernelGGL(norm_kernel<true>, dim3((rows + 3) / 4), dim3(256), 0, s, x, gamma, (const bf16_t*)nullptr, y, rows,
                     cols, eps, row_index, 0);
  return hipGetLastError();
}
