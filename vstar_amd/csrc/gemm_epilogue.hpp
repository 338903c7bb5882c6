// gemm_epilogue.hpp — shared GEMM epilogue: one 16x16 accumulator fragment slice (4 consecutive output columns of
// one row) -> bias, activation with the reference's bf16 rounding points, residual, bf16x4 / f32x4 store.
#pragma once
#include "common.hpp"
#include "kernels.hpp"

__device__ __forceinline__ int64_t gemm_map_row(int r, int group, int64_t gstride, int64_t off) {
  if (group <= 0) return (int64_t)r;
  int g = r / group;
  return (int64_t)g * gstride + off + (r - g * group);
}

// a = accumulator (gate accumulator for SILU_MUL), u = up accumulator (SILU_MUL only); col = first output column
template <int EPI, bool OUT_F32>
__device__ __forceinline__ void gemm_epilogue_store(const GemmParams& p, int64_t crow, int col, int n_out, const f32x4& a,
                                                    const f32x4& u) {
  if (col >= n_out) return;
  const bool full = (col + 3 < n_out);
  float o[4];
  if (EPI == VSTAR_EPI_SILU_MUL) {
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = act_silu_bf16(rbf(a[e])) * rbf(u[e]);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = a[e];
    if (p.bias) {
      if (full) {
        const bf16x4 b = *(const bf16x4*)(p.bias + col);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] += bf2f((bf16_t)b[e]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (col + e < n_out) o[e] += bf2f(p.bias[col + e]);
      }
    }
    if (!OUT_F32) {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = rbf(o[e]);       // nn.Linear output is bf16 in the reference
    }
    if (EPI == VSTAR_EPI_QUICK_GELU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = OUT_F32 ? o[e] / (1.0f + __expf(-1.702f * o[e])) : act_quick_gelu_bf16(o[e]);
    } else if (EPI == VSTAR_EPI_GELU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = act_gelu_erf(o[e]);
    } else if (EPI == VSTAR_EPI_RELU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
    }
    if (p.res) {
      const bf16_t* rp = p.res + crow * p.ldr + col;
      if (!OUT_F32) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = rbf(o[e]);     // activation output rounded before the add
      }
      if (full) {
        const bf16x4 rv = *(const bf16x4*)rp;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] += bf2f((bf16_t)rv[e]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (col + e < n_out) o[e] += bf2f(rp[e]);
      }
    }
  }
  if (OUT_F32) {
    float* c = (float*)p.C + crow * p.ldc + col;
    if (full && ((((uintptr_t)c) & 15) == 0)) {
      *(f32x4*)c = (f32x4){o[0], o[1], o[2], o[3]};
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) if (col + e < n_out) c[e] = o[e];
    }
  } else {
    bf16_t* c = (bf16_t*)p.C + crow * p.ldc + col;
    if (full && ((((uintptr_t)c) & 7) == 0)) {
      bf16x4 v = {(short)f2bf(o[0]), (short)f2bf(o[1]), (short)f2bf(o[2]), (short)f2bf(o[3])};
      *(bf16x4*)c = v;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) if (col + e < n_out) c[e] = f2bf(o[e]);
    }
  }
}
