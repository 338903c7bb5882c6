// gemm_epilogue.hpp — shared GEMM epilogue: one 16x16 accumulator fragment slice (4 consecutive output columns of
// one row) -> bias, activation with the reference's bf16 rounding points, residual, lpx4 / f32x4 store.
#pragma once
#include "common.hpp"
#include "kernels.hpp"

namespace VS_NS {

__device__ __forceinline__ int64_t gemm_map_row(int r, int group, int64_t gstride, int64_t off) {
  if (group <= 0) return (int64_t)r;
  int g = r / group;
  return (int64_t)g * gstride + off + (r - g * group);
}

// Stage 1 of the epilogue, in the MFMA accumulator layout: bias + activation with the reference's bf16 rounding points
// (no residual).  a = accumulator (gate accumulator for SILU_MUL), u = up accumulator; col = first output column.
template <int EPI, bool OUT_F32>
__device__ __forceinline__ void gemm_epilogue_values(const GemmParams& p, int col, int n_out, const f32x4& a, const f32x4& u,
                                                     float (&o)[4]) {
  const bool full = (col + 3 < n_out);
  if (EPI == VSTAR_EPI_SILU_MUL) {
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = act_silu_bf16(rlp(a[e])) * rlp(u[e]);
    return;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = a[e];
  if (p.bias) {
    if (full) {
      const lpx4 b = *(const lpx4*)(p.bias + col);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] += lp2f((lp_t)b[e]);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) if (col + e < n_out) o[e] += lp2f(p.bias[col + e]);
    }
  }
  if (!OUT_F32) {
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = rlp(o[e]);       // nn.Linear output is bf16 in the reference
  }
  if (EPI == VSTAR_EPI_QUICK_GELU) {
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = OUT_F32 ? o[e] * fast_sigmoid(1.702f * o[e]) : act_quick_gelu_bf16(o[e]);
  } else if (EPI == VSTAR_EPI_GELU) {
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = act_gelu_erf(o[e]);
  } else if (EPI == VSTAR_EPI_RELU) {
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
  }
}

// Direct epilogue (accumulator layout -> global): 4 consecutive output columns of one row per call.
template <int EPI, bool OUT_F32>
__device__ __forceinline__ void gemm_epilogue_store(const GemmParams& p, int64_t crow, int col, int n_out, const f32x4& a,
                                                    const f32x4& u, float* stored = nullptr) {
  if (stored) stored[0] = stored[1] = stored[2] = stored[3] = 0.f;
  if (col >= n_out) return;
  const bool full = (col + 3 < n_out);
  float o[4];
  gemm_epilogue_values<EPI, OUT_F32>(p, col, n_out, a, u, o);
  if (p.res) {
    const lp_t* rp = p.res + crow * p.ldr + col;
    if (!OUT_F32) {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = rlp(o[e]);     // activation output rounded before the add
    }
    if (full) {
      const lpx4 rv = *(const lpx4*)rp;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] += lp2f((lp_t)rv[e]);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) if (col + e < n_out) o[e] += lp2f(rp[e]);
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
    lp_t* c = (lp_t*)p.C + crow * p.ldc + col;
    if (stored) {
#pragma unroll
      for (int e = 0; e < 4; ++e) stored[e] = (col + e < n_out) ? rlp(o[e]) : 0.f;       // what lands in memory
    }
    if (full && ((((uintptr_t)c) & 7) == 0)) {
      lpx4 v = {(short)f2lp(o[0]), (short)f2lp(o[1]), (short)f2lp(o[2]), (short)f2lp(o[3])};
      *(lpx4*)c = v;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) if (col + e < n_out) c[e] = f2lp(o[e]);
    }
  }
}

// Stage 2 of the LDS-transposed bf16 epilogue: 8 consecutive output columns of one row, holding bf16(acc + bias)
// (or the finished SiLU*up product) -> activation -> + residual -> one 16-byte store.  Whole 128-byte lines per 8 lanes
// instead of 32-byte fragments.  The transcendental activations live here (static 8-element bodies) so that the
// accumulator-indexed stage-1 loops stay small enough to unroll (a runtime-indexed acc[] would go to scratch).
// the activation of 8 packed values after the transpose (QUICK_GELU / GELU / RELU; NONE and SILU_MUL have nothing left to do here)
template <int EPI>
__device__ __forceinline__ void gemm_epilogue_act8(lpx8& v) {
  if (EPI == VSTAR_EPI_QUICK_GELU) {
#if !defined(VSTAR_LP_F16) && !defined(VSTAR_EXACT_SIGMOID) && !defined(VSTAR_SCALAR_GELU)
    // two neighbouring elements per step on the packed fp32 pipes: the operations, constants and rounding points of
    // act_quick_gelu_bf16 (u = round(1.702 t); s = round(rcp(1 + exp(-u))); out = round(t s)) — only v_exp_f32 / v_rcp_f32 stay scalar
    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
    typedef __attribute__((ext_vector_type(2))) float f2_t;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2_t;
    u32x4 w = __builtin_bit_cast(u32x4, v);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f2_t t = {__uint_as_float(w[k] << 16), __uint_as_float(w[k] & 0xffff0000u)};
      const uint32_t ub = __builtin_bit_cast(uint32_t, __builtin_convertvector(t * 1.702f, bf2_t));
      const f2_t u = {__uint_as_float(ub << 16), __uint_as_float(ub & 0xffff0000u)};
      f2_t s = {__builtin_amdgcn_rcpf(1.0f + __expf(-u[0])), __builtin_amdgcn_rcpf(1.0f + __expf(-u[1]))};
      const uint32_t sb = __builtin_bit_cast(uint32_t, __builtin_convertvector(s, bf2_t));
      s = (f2_t){__uint_as_float(sb << 16), __uint_as_float(sb & 0xffff0000u)};
      w[k] = __builtin_bit_cast(uint32_t, __builtin_convertvector(t * s, bf2_t));
    }
    v = __builtin_bit_cast(lpx8, w);
#else
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (short)f2lp(act_quick_gelu_bf16(lp2f((lp_t)v[e])));
#endif
  } else if (EPI == VSTAR_EPI_GELU) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (short)f2lp(act_gelu_erf(lp2f((lp_t)v[e])));
  } else if (EPI == VSTAR_EPI_RELU) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (short)f2lp(fmaxf(lp2f((lp_t)v[e]), 0.f));
  }
}

// `res_done`: the caller has already added the residual (gemm256 adds it in the accumulator layout, see there).
template <int EPI>
__device__ __forceinline__ void gemm_epilogue_store_row8(const GemmParams& p, int64_t crow, int col, int n_out, lpx8& v,
                                                         bool res_done = false) {
  if (col >= n_out) return;
  gemm_epilogue_act8<EPI>(v);
  lp_t* c = (lp_t*)p.C + crow * p.ldc + col;
  const bool full = (col + 7 < n_out) && ((((uintptr_t)c) & 15) == 0);
  if (p.res && !res_done) {
    const lp_t* rp = p.res + crow * p.ldr + col;
    if (full && ((((uintptr_t)rp) & 15) == 0)) {
      const lpx8 rv = *(const lpx8*)rp;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (short)f2lp(lp2f((lp_t)v[e]) + lp2f((lp_t)rv[e]));
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) if (col + e < n_out) v[e] = (short)f2lp(lp2f((lp_t)v[e]) + lp2f(rp[e]));
    }
  }
  if (full) {
    // streaming store, except when the rows are the next linear's A operand right away (sumsq_out = the residual stream feeding a
    // folded RMSNorm): those should stay cache-resident
    if (p.sumsq_out) *(lpx8*)c = v;
    else __builtin_nontemporal_store(v, (lpx8*)c);
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) if (col + e < n_out) c[e] = (lp_t)v[e];
  }
}

// GemmParams::sumsq_out: sum of squares of a 64-column span of STORED values, canonical order (see kernels.hpp / norm.hip):
//   chunk (8 consecutive columns): (c0^2 + c1^2 + c2^2 + c3^2 sequentially) + (c4..c7 likewise); chunk pairs; 16-col fragments pairwise.
// gemm256 layout: 8 consecutive lanes hold the 8 chunks of the span.
__device__ __forceinline__ float gemm_sumsq_span64_chunks(const lpx8& v) {
  float f[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) f[e] = lp2f((lp_t)v[e]);
  const float h0 = ((f[0] * f[0] + f[1] * f[1]) + f[2] * f[2]) + f[3] * f[3];
  const float h1 = ((f[4] * f[4] + f[5] * f[5]) + f[6] * f[6]) + f[7] * f[7];
  return dpp_add_half_mirror(dpp_add_xor2(dpp_add_xor1(h0 + h1)));
}
// GemmParams::stats_sum: the plain SUM of the same span, same tree (the second statistic of a LayerNorm folded into the consumer)
__device__ __forceinline__ float gemm_sum_span64_chunks(const lpx8& v) {
  float f[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) f[e] = lp2f((lp_t)v[e]);
  const float h0 = ((f[0] + f[1]) + f[2]) + f[3];
  const float h1 = ((f[4] + f[5]) + f[6]) + f[7];
  return dpp_add_half_mirror(dpp_add_xor2(dpp_add_xor1(h0 + h1)));
}
// gemm128 layout (accumulator layout): lane (fr = lane & 15, fq = lane >> 4) holds columns n*16 + fq*4 + {0..3} of fragment n = 0..3
// as o[n][0..3]; half-chunks meet across lane ^ 16, chunks of a fragment across lane ^ 32, fragments inside the lane.
__device__ __forceinline__ float gemm_sumsq_span64_frags(const float (&o)[4][4]) {
  float pn[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    float h = ((o[n][0] * o[n][0] + o[n][1] * o[n][1]) + o[n][2] * o[n][2]) + o[n][3] * o[n][3];
    h += __shfl_xor(h, 16, 64);      // h0 + h1 of the chunk
    h += __shfl_xor(h, 32, 64);      // the fragment's two chunks
    pn[n] = h;
  }
  return (pn[0] + pn[1]) + (pn[2] + pn[3]);
}
__device__ __forceinline__ float gemm_sum_span64_frags(const float (&o)[4][4]) {
  float pn[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    float h = ((o[n][0] + o[n][1]) + o[n][2]) + o[n][3];
    h += __shfl_xor(h, 16, 64);
    h += __shfl_xor(h, 32, 64);
    pn[n] = h;
  }
  return (pn[0] + pn[1]) + (pn[2] + pn[3]);
}

}  // namespace VS_NS
