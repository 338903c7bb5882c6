// quant.hip — W8A8 support for BASELINE config 5 (fp8 weights on the CDNA4 fp8 MFMA): symmetric per-row quantisation of a
// 16-bit matrix to OCP fp8 e4m3 (scale = absmax / 448, round-to-nearest-even by v_cvt_pk_fp8_f32), used for
//   * weights, once at load time: rows of the packed [Npad, K] matrix = output channels  -> per-output-channel scales
//   * activations, per call: rows = tokens                                               -> per-token scales
// and a LlamaRMSNorm variant that emits the quantised row directly (the norm already holds the row in registers).
// The reference has no fp8 path; the oracle's fake-quant restatement (oracle/vsm_oracle.py: fp8 helpers) is the parity target.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "common.hpp"
#include "kernels.hpp"
#include "mx.hpp"

namespace VS_NS {

namespace {

constexpr float FP8_MAX = 448.0f;

__device__ __forceinline__ uint32_t pack4_fp8(float a, float b, float c, float d) {
  uint32_t v = 0;
  v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, v, false);
  v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
  return v;
}

// one workgroup (256 threads) per row; cols % 8 == 0
__global__ __launch_bounds__(256) void quantize_rows_kernel(const lp_t* __restrict__ x, int64_t ldx, uint8_t* __restrict__ q,
                                                            int64_t ldq, float* __restrict__ scale, int cols) {
  __shared__ float red[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  const lp_t* xr = x + (int64_t)row * ldx;
  float mx = 0.f;
  for (int c = tid * 8; c < cols; c += 256 * 8) {
    const lpx8 v = *(const lpx8*)(xr + c);
#pragma unroll
    for (int e = 0; e < 8; ++e) mx = fmaxf(mx, fabsf(lp2f((lp_t)v[e])));
  }
  mx = wave_max(mx);
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float sc = mx > 0.f ? mx / FP8_MAX : 1.0f;
  const float inv = 1.0f / sc;
  if (tid == 0) scale[row] = sc;
  uint8_t* qr = q + (int64_t)row * ldq;
  for (int c = tid * 8; c < cols; c += 256 * 8) {
    const lpx8 v = *(const lpx8*)(xr + c);
    uint2 o;
    o.x = pack4_fp8(lp2f((lp_t)v[0]) * inv, lp2f((lp_t)v[1]) * inv, lp2f((lp_t)v[2]) * inv, lp2f((lp_t)v[3]) * inv);
    o.y = pack4_fp8(lp2f((lp_t)v[4]) * inv, lp2f((lp_t)v[5]) * inv, lp2f((lp_t)v[6]) * inv, lp2f((lp_t)v[7]) * inv);
    *(uint2*)(qr + c) = o;
  }
}

// LlamaRMSNorm (same arithmetic and rounding points as norm_kernel<true>) + per-row fp8 quantisation of its 16-bit output.
// One wave per row, cols <= 4096.
__global__ __launch_bounds__(256) void rmsnorm_quant_kernel(const lp_t* __restrict__ x, const lp_t* __restrict__ gamma,
                                                            uint8_t* __restrict__ q, float* __restrict__ scale, int rows, int cols,
                                                            float eps) {
  constexpr int MAXCH = 8;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const lp_t* xr = x + (int64_t)row * cols;
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
        sum += v[c][e] * v[c][e];
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[c][e] = 0.f;
    }
  }
  sum = wave_sum(sum);
  const float rstd = rsqrtf(sum / (float)cols + eps);
  float mx = 0.f;
#pragma unroll
  for (int c = 0; c < MAXCH; ++c) {
    const int vi = c * 64 + lane;
    if (vi < nvec) {
      const lpx8 g = *(const lpx8*)(gamma + vi * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[c][e] = rlp(lp2f((lp_t)g[e]) * rlp(v[c][e] * rstd));     // the 16-bit value the plain norm kernel would store
        mx = fmaxf(mx, fabsf(v[c][e]));
      }
    }
  }
  mx = wave_max(mx);
  const float sc = mx > 0.f ? mx / FP8_MAX : 1.0f;
  const float inv = 1.0f / sc;
  if (lane == 0) scale[row] = sc;
  uint8_t* qr = q + (int64_t)row * cols;
#pragma unroll
  for (int c = 0; c < MAXCH; ++c) {
    const int vi = c * 64 + lane;
    if (vi < nvec) {
      uint2 o;
      o.x = pack4_fp8(v[c][0] * inv, v[c][1] * inv, v[c][2] * inv, v[c][3] * inv);
      o.y = pack4_fp8(v[c][4] * inv, v[c][5] * inv, v[c][6] * inv, v[c][7] * inv);
      *(uint2*)(qr + vi * 8) = o;
    }
  }
}

// block-scaled quantisation (mx.hpp): one thread per 8 values, four threads per block of 32
__global__ __launch_bounds__(256) void quantize_rows_mx_kernel(const lp_t* __restrict__ x, int64_t ldx, uint8_t* __restrict__ q, int64_t ldq,
                                                               uint8_t* __restrict__ scales, int rows, int cols) {
  const int per_row = cols >> 3;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)rows * per_row) return;            // (per_row % 4 == 0: a block's four threads leave together)
  const int row = (int)(idx / per_row), c8 = (int)(idx - (int64_t)row * per_row);
  const lpx8 v = *(const lpx8*)(x + (int64_t)row * ldx + c8 * 8);
  float f[8], mx = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    f[e] = lp2f((lp_t)v[e]);
    mx = fmaxf(mx, fabsf(f[e]));
  }
  mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
  const uint32_t e8 = mx_e8m0(mx);
  const float inv = mx_inv_scale(e8);
  uint2 o;
  o.x = mx_pack4(f[0] * inv, f[1] * inv, f[2] * inv, f[3] * inv);
  o.y = mx_pack4(f[4] * inv, f[5] * inv, f[6] * inv, f[7] * inv);
  *(uint2*)(q + (int64_t)row * ldq + c8 * 8) = o;
  if ((c8 & 3) == 0) scales[mx_scale_offset(row, c8 >> 2, rows >> 7)] = (uint8_t)e8;
}

}  // namespace

hipError_t quantize_rows_mx(const lp_t* x, int64_t ldx, uint8_t* q, int64_t ldq, uint8_t* scales, int rows, int cols, hipStream_t s) {
  if (rows <= 0) return hipSuccess;
  if (rows % 128 || cols % 128 || ldx % 8 || ldq % 8) return hipErrorInvalidValue;
  const int64_t n = (int64_t)rows * (cols / 8);
  hipLaunchKernelGGL(quantize_rows_mx_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, ldx, q, ldq, scales, rows, cols);
  return hipGetLastError();
}

hipError_t quantize_rows_fp8(const lp_t* x, int64_t ldx, uint8_t* q, int64_t ldq, float* scale, int rows, int cols,
                             hipStream_t s) {
  if (rows <= 0) return hipSuccess;
  if (cols % 8 || ldx % 8 || ldq % 8) return hipErrorInvalidValue;
  hipLaunchKernelGGL(quantize_rows_kernel, dim3(rows), dim3(256), 0, s, x, ldx, q, ldq, scale, cols);
  return hipGetLastError();
}

hipError_t rmsnorm_quant_fp8(const lp_t* x, const lp_t* gamma, uint8_t* q, float* scale, int rows, int cols, float eps,
                             hipStream_t s) {
  if (rows <= 0) return hipSuccess;
  if (cols % 8 || cols > 4096) return hipErrorInvalidValue;
  hipLaunchKernelGGL(rmsnorm_quant_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, gamma, q, scale, rows, cols, eps);
  return hipGetLastError();
}

}  // namespace VS_NS
