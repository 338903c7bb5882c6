// gemm256_direct_epilogue.hpp — the in-register ("direct") epilogue of the 256 x 256 GEMM kernels, shared by the 8-wave gemm256.hip
// and the 4-wave gemm4w.hip (round 6: one copy instead of two).
//
// Unit of work: ONE virtual wave (wr, wc) of gemm256's 2 x 4 wave grid = 128 rows x 64 W rows of a tile, accumulators in the MFMA
// layout: lane (fr = lane & 15, fq = lane >> 4) holds, for row group m (rows 16 m + fr) and fragment n, the four columns
// n * 16 + fq * 4 + e.  The W rows of a direct tile were DMA'd in a PERMUTED order (the DMA source address is per lane, so which W row
// an LDS row holds is free):  LDS row (n, i)  <-  W row (n >> 1) * 32 + (i >> 2) * 8 + (n & 1) * 4 + (i & 3)
// so that the lane's fragments 0|1 are 8 CONSECUTIVE output columns (16 bytes) at fq * 8 and fragments 2|3 the 8 at 32 + fq * 8; RoPE
// tiles (a head = 128 columns = the two neighbouring virtual waves) instead take d = (wc & 1) * 32 + fq * 8 .. and d + 64 into ONE lane,
// the rotate-half partner without any exchange; SiLU tiles (storage rows: 16 gate rows, 16 up rows, ...) put gate / up of 8
// consecutive outputs into fragments 0,2 / 1,3.  Bias / activation / residual / RoPE / statistics and two 16-byte stores per row then
// happen in registers — no LDS slab, no transpose.  Same dot products, k order, rounding points and statistics tree as the
// LDS-transposed epilogue of gemm256.hip (edge tiles) and as gemm128: bit-identical.
//
// `acc_row(std::integral_constant<int, m>, f32x4 (&a)[4])` hands over the FINAL fp32 accumulators of row group m (dequantised /
// row-scaled by the caller): gemm256 copies them out of its register array, gemm4w reads them from the AGPRs sixteen at a time.
#pragma once
#include "common.hpp"
#include "kernels.hpp"
#include "gemm_epilogue.hpp"
#include <type_traits>
#include <utility>

namespace VS_NS {

template <int... I, class F>
__device__ __forceinline__ void gemm_static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void gemm_static_for(F&& f) { gemm_static_for_impl(std::make_integer_sequence<int, N>{}, f); }

// STATS: whether GemmParams::sumsq_out / stats_sum are honoured (not in the W8A8 instantiations)
// ROWGUARD (gemm4w, whose last row tile may be ragged — gemm256 sends its edge tiles through the LDS-transposed epilogue): rows >= M
// are computed like any other (their A rows read as zero) but neither stored nor read — residual, RoPE position and statistics
// accesses of such a row are skipped lane by lane.
template <int EPI, bool STATS, bool ROWGUARD = false, class AccRow>
__device__ __forceinline__ void gemm256_direct_epilogue(const GemmParams& p, int em0, int en0, int wr, int wc, int fr, int fq, AccRow&& acc_row) {
  constexpr bool SILU = (EPI == VSTAR_EPI_SILU_MUL);
  bool rope_tile = false;
  if constexpr (EPI == VSTAR_EPI_NONE) rope_tile = p.rope_cs != nullptr && en0 < p.rope_cols;
  int col_a, col_b;
  if (SILU) { col_a = (en0 + wc * 64) / 2 + fq * 8; col_b = col_a; }
  else if (rope_tile) { col_a = en0 + (wc >> 1) * 128 + (wc & 1) * 32 + fq * 8; col_b = col_a + 64; }
  else { col_a = en0 + wc * 64 + fq * 8; col_b = col_a + 32; }
  const int row0 = em0 + wr * 128 + fr;
  lp_t* crow = (lp_t*)p.C + (int64_t)row0 * p.ldc;
  if constexpr (SILU) {
    gemm_static_for<8>([&](auto mc) {
      f32x4 a[4];
      acc_row(mc, a);
      lpx8 v;
#if !defined(VSTAR_LP_F16) && !defined(VSTAR_EXACT_SIGMOID) && !defined(VSTAR_SCALAR_GELU)
      {   // two neighbouring outputs per step on the packed fp32 pipes; operations and rounding points of the scalar form below
        typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
        typedef __attribute__((ext_vector_type(2))) float f2_t;
        typedef __attribute__((ext_vector_type(2))) __bf16 bf2_t;
        auto rnd = [](f2_t x) {
          const uint32_t b = __builtin_bit_cast(uint32_t, __builtin_convertvector(x, bf2_t));
          return (f2_t){__uint_as_float(b << 16), __uint_as_float(b & 0xffff0000u)};
        };
        u32x4 w;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int n = (k >> 1) * 2, e = (k & 1) * 2;
          const f2_t g = rnd((f2_t){a[n][e], a[n][e + 1]}), u = rnd((f2_t){a[n + 1][e], a[n + 1][e + 1]});
          const f2_t sg = {__builtin_amdgcn_rcpf(1.0f + __expf(-g[0])), __builtin_amdgcn_rcpf(1.0f + __expf(-g[1]))};
          w[k] = __builtin_bit_cast(uint32_t, __builtin_convertvector(rnd(g * sg) * u, bf2_t));
        }
        v = __builtin_bit_cast(lpx8, w);
      }
#else
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = (short)f2lp(act_silu_bf16(rlp(a[0][e])) * rlp(a[1][e]));
        v[4 + e] = (short)f2lp(act_silu_bf16(rlp(a[2][e])) * rlp(a[3][e]));
      }
#endif
      if (!ROWGUARD || row0 + decltype(mc)::value * 16 < p.M) __builtin_nontemporal_store(v, (lpx8*)(crow + col_a));
      crow += 16 * p.ldc;
    });
  } else {
    lpx8 pa[8], pb[8];
    {
      float bia[8], bib[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) bia[e] = bib[e] = 0.f;
      if (p.bias) {
        const lpx8 b0 = *(const lpx8*)(p.bias + col_a), b1 = *(const lpx8*)(p.bias + col_b);
#pragma unroll
        for (int e = 0; e < 8; ++e) { bia[e] = lp2f((lp_t)b0[e]); bib[e] = lp2f((lp_t)b1[e]); }
      }
      if (p.bias) {
        gemm_static_for<8>([&](auto mc) {
          constexpr int m = decltype(mc)::value;
          f32x4 a[4];
          acc_row(mc, a);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            pa[m][e] = (short)f2lp(a[0][e] + bia[e]);
            pa[m][4 + e] = (short)f2lp(a[1][e] + bia[4 + e]);
            pb[m][e] = (short)f2lp(a[2][e] + bib[e]);
            pb[m][4 + e] = (short)f2lp(a[3][e] + bib[4 + e]);
          }
        });
      } else {
        // no bias (every LLaMA linear): 256 adds of +0.0f less per lane on a path that is VALU-bound with one wave per SIMD
        // (tools/gemm4w_timeline.py: 3.9 us of epilogue arithmetic per tile).  Same bits: an MFMA chain that starts from +0 never
        // yields -0, the only value x + 0.0f would change.
        gemm_static_for<8>([&](auto mc) {
          constexpr int m = decltype(mc)::value;
          f32x4 a[4];
          acc_row(mc, a);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            pa[m][e] = (short)f2lp(a[0][e]);
            pa[m][4 + e] = (short)f2lp(a[1][e]);
            pb[m][e] = (short)f2lp(a[2][e]);
            pb[m][4 + e] = (short)f2lp(a[3][e]);
          }
        });
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (EPI == VSTAR_EPI_NONE) {
      if (rope_tile) {
        const int rope_d = (wc & 1) * 32 + fq * 8;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          const int row = (!ROWGUARD || row0 + m * 16 < p.M) ? row0 + m * 16 : row0 & 15;      // (ragged tail: any valid row, result unused)
          int pos = row % p.rope_S;
          if (p.rope_R0 > 0 && pos >= p.rope_R0) pos = p.rope_Lc + ((pos - p.rope_R0) & 31);
          if (p.rope_tail > 0) pos = row >= p.rope_tail ? row - p.rope_tail : pos + p.rope_pos0;
          const lpx8 c8 = *(const lpx8*)(p.rope_cs + (int64_t)pos * 128 + rope_d);
          const lpx8 s8 = *(const lpx8*)(p.rope_cs + (int64_t)pos * 128 + 64 + rope_d);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float c = lp2f((lp_t)c8[e]), sn = lp2f((lp_t)s8[e]);
            const float xa = lp2f((lp_t)pa[m][e]), xb = lp2f((lp_t)pb[m][e]);
            pa[m][e] = (short)f2lp(rlp(xa * c) + rlp(-1.0f * xb * sn));
            pb[m][e] = (short)f2lp(rlp(xb * c) + rlp(1.0f * xa * sn));
          }
        }
      }
    } else {
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        gemm_epilogue_act8<EPI>(pa[m]);
        gemm_epilogue_act8<EPI>(pb[m]);
      }
    }
    if (p.res) {
      __builtin_amdgcn_sched_barrier(0);
      lpx8 ra[8], rb[8];
      const lp_t* rrow = p.res + (int64_t)row0 * p.ldr;
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        if (!ROWGUARD || row0 + m * 16 < p.M) {
          ra[m] = *(const lpx8*)(rrow + col_a);
          rb[m] = *(const lpx8*)(rrow + col_b);
        } else {
          ra[m] = rb[m] = (lpx8){0, 0, 0, 0, 0, 0, 0, 0};
        }
        rrow += 16 * p.ldr;
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          pa[m][e] = (short)f2lp(lp2f((lp_t)pa[m][e]) + lp2f((lp_t)ra[m][e]));
          pb[m][e] = (short)f2lp(lp2f((lp_t)pb[m][e]) + lp2f((lp_t)rb[m][e]));
        }
    }
    float* sq = nullptr;
    if constexpr (EPI == VSTAR_EPI_NONE && STATS) {
      if (p.sumsq_out) sq = p.sumsq_out + (int64_t)row0 * p.sumsq_ld + (en0 + wc * 64) / 64;
    }
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      __builtin_amdgcn_sched_barrier(0);
      const bool row_ok = !ROWGUARD || row0 + m * 16 < p.M;
      if (!row_ok) {
      } else if (p.sumsq_out) {          // the rows are the next linear's A operand right away: keep them cache-resident
        *(lpx8*)(crow + col_a) = pa[m];
        *(lpx8*)(crow + col_b) = pb[m];
      } else {
        __builtin_nontemporal_store(pa[m], (lpx8*)(crow + col_a));
        __builtin_nontemporal_store(pb[m], (lpx8*)(crow + col_b));
      }
      crow += 16 * p.ldc;
      if constexpr (EPI == VSTAR_EPI_NONE && STATS) {
        if (sq) {
          float fa[8], fb[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) { fa[e] = lp2f((lp_t)pa[m][e]); fb[e] = lp2f((lp_t)pb[m][e]); }
          float qa = (((fa[0] * fa[0] + fa[1] * fa[1]) + fa[2] * fa[2]) + fa[3] * fa[3]) +
                     (((fa[4] * fa[4] + fa[5] * fa[5]) + fa[6] * fa[6]) + fa[7] * fa[7]);
          float qb = (((fb[0] * fb[0] + fb[1] * fb[1]) + fb[2] * fb[2]) + fb[3] * fb[3]) +
                     (((fb[4] * fb[4] + fb[5] * fb[5]) + fb[6] * fb[6]) + fb[7] * fb[7]);
          qa += __shfl_xor(qa, 16, 64); qa += __shfl_xor(qa, 32, 64);
          qb += __shfl_xor(qb, 16, 64); qb += __shfl_xor(qb, 32, 64);
          if (fq == 0 && row_ok) sq[0] = qa + qb;
          if (p.stats_sum) {
            float sa = (((fa[0] + fa[1]) + fa[2]) + fa[3]) + (((fa[4] + fa[5]) + fa[6]) + fa[7]);
            float sb = (((fb[0] + fb[1]) + fb[2]) + fb[3]) + (((fb[4] + fb[5]) + fb[6]) + fb[7]);
            sa += __shfl_xor(sa, 16, 64); sa += __shfl_xor(sa, 32, 64);
            sb += __shfl_xor(sb, 16, 64); sb += __shfl_xor(sb, 32, 64);
            if (fq == 0 && row_ok) sq[p.stats_sum] = sa + sb;
          }
          sq += 16 * p.sumsq_ld;
        }
      }
    }
  }
}

}  // namespace VS_NS
