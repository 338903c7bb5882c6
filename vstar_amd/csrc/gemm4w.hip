// gemm4w.hip — the 256 x 256 x 64 GEMM tile with FOUR waves (one per SIMD), each owning a 128 x 128 output in 256 AGPR
// accumulators, and a hand-scheduled K loop (tools/gen_gemm4w_asm.py -> gemm4w_loop.inc).  Round 6.
//
// Why a second 256^2 kernel.  On real operands the GEMM family runs ON the board's 1400 W cap: throughput is set by
// energy per FLOP.  The 8-wave gemm256 reads 0.375 fragment ds_read_b128 per MFMA (128 x 64 per wave) and pays four barrier
// rendezvous per K-tile between two wave groups; this kernel reads 0.25 (128 x 128 per wave), keeps all 512 registers of a SIMD
// for ONE wave (256 accumulators in AGPRs + two fragment sets), and hides every LDS read, DMA issue and barrier between the
// MFMAs of that wave.  The structure — 4 waves, AGPR accumulators, LDS-DMA for both operands, the two halves of an LDS buffer
// released separately so that the DMA of K-tile T+2 starts in the first half of K-tile T — is what the vendor's own
// 256x256x64 kernel does (facts read from its code object: DESIGN.md §5.1); the code is ours: same LDS image, same DMA
// addressing, same permuted-W in-register epilogue and same k order as gemm256.hip, hence bit-identical results.
//
// Domain (everything else stays on gemm256.hip): bf16 / fp16 operands and output, M % 256 == 0 — or any M >= 1024 when K >= 4096: the last
// row tile may be ragged (template parameter RG) —, N % 256 == 0, K % 128 == 0, identity row maps, 16-byte aligned C / residual / bias —
// i.e. launches whose tiles all take the direct epilogue;
// epilogues NONE (+bias, +residual, fused RoPE, folded-norm row scale, sum-of-squares / sum statistics), QUICK_GELU, RELU, SILU_MUL.
#include "common.hpp"
#include "kernels.hpp"
#include "gemm_epilogue.hpp"
#include "gemm256_direct_epilogue.hpp"
#include "mx.hpp"
#include <type_traits>
#include <utility>

#ifdef VSTAR_LP_F16
#define G4W_DT "f16"
#else
#define G4W_DT "bf16"
#endif
#ifndef G4W_LOOP_INC        // variant builds (tools/build_variant.sh): another generated schedule
#define G4W_LOOP_INC "gemm4w_loop.inc"
#endif
#include G4W_LOOP_INC

namespace VS_NS {

namespace {

constexpr int BM = 256, BN = 256;
constexpr int LDS_TOTAL = 2 * 65536;

// The accumulators sit in a0..a255 BEHIND the compiler's back: they are clobbers of the K-loop statement, so to the register allocator
// the AGPRs are free afterwards — and under pressure it parks VGPR values there (round 6: the W piece offsets landed on a2..a9
// before the epilogue had read them).  Every read statement therefore clobbers the whole AGPR file: no compiler value can live in an
// AGPR across any of them, i.e. anywhere between the K loop and the last read; tools/check_gemm4w_agpr.py (run by the CPU test suite
// on the compiler's assembly) proves that no compiler-written AGPR instruction remains in that window.
#define G4W_A16(b) "a" #b "0", "a" #b "1", "a" #b "2", "a" #b "3", "a" #b "4", "a" #b "5", "a" #b "6", "a" #b "7", "a" #b "8", "a" #b "9"
#define G4W_ALL_AGPRS                                                                                                              \
  "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", G4W_A16(1), G4W_A16(2), G4W_A16(3), G4W_A16(4), G4W_A16(5), G4W_A16(6), \
  G4W_A16(7), G4W_A16(8), G4W_A16(9), G4W_A16(10), G4W_A16(11), G4W_A16(12), G4W_A16(13), G4W_A16(14), G4W_A16(15), G4W_A16(16),     \
  G4W_A16(17), G4W_A16(18), G4W_A16(19), G4W_A16(20), G4W_A16(21), G4W_A16(22), G4W_A16(23), G4W_A16(24), "a250", "a251", "a252",   \
  "a253", "a254", "a255"
#ifdef G4W_TIMELINE   // diagnostic builds only (tools/gemm4w_timeline.py): wall-clock stamps (100 MHz) of the tile phases, wave 0 of WG 0 / 100
__device__ unsigned long long g4w_timeline[2][64][8];
#define G4W_STAMP(pt)                                                                                         \
  do {                                                                                                        \
    if ((threadIdx.x & 63) == 0 && wave == 0 && (blockIdx.x == 0 || blockIdx.x == 100) && tl_tile < 64)        \
      g4w_timeline[blockIdx.x ? 1 : 0][tl_tile][pt] = wall_clock64();                                         \
  } while (0)
#else
#define G4W_STAMP(pt) do { } while (0)
#endif

// fragment (m, n') of the wave's 8 x 8 lives in a[(m * 8 + n') * 4 ..+3] (tools/gen_gemm4w_asm.py::mfma).  Row m of the 128 x 64 half H of
// the wave's tile = fragments n' = 4 H .. 4 H + 3 = SIXTEEN CONSECUTIVE AGPRs a[(m * 8 + 4 H) * 4 ..+15], read by ONE statement: the
// compiler puts an `s_nop` behind every inline-asm statement it cannot see into — one statement per register was 256 of them per tile.
template <int H, int M>
__device__ __forceinline__ void load_acc_row(f32x4 (&a)[4]) {
  constexpr int B = (M * 8 + H * 4) * 4;
  asm volatile("v_accvgpr_read_b32 %0, a[%c16]\n\tv_accvgpr_read_b32 %1, a[%c16+1]\n\tv_accvgpr_read_b32 %2, a[%c16+2]\n\tv_accvgpr_read_b32 %3, a[%c16+3]\n\t"
               "v_accvgpr_read_b32 %4, a[%c16+4]\n\tv_accvgpr_read_b32 %5, a[%c16+5]\n\tv_accvgpr_read_b32 %6, a[%c16+6]\n\tv_accvgpr_read_b32 %7, a[%c16+7]\n\t"
               "v_accvgpr_read_b32 %8, a[%c16+8]\n\tv_accvgpr_read_b32 %9, a[%c16+9]\n\tv_accvgpr_read_b32 %10, a[%c16+10]\n\tv_accvgpr_read_b32 %11, a[%c16+11]\n\t"
               "v_accvgpr_read_b32 %12, a[%c16+12]\n\tv_accvgpr_read_b32 %13, a[%c16+13]\n\tv_accvgpr_read_b32 %14, a[%c16+14]\n\tv_accvgpr_read_b32 %15, a[%c16+15]"
               : "=v"(a[0][0]), "=v"(a[0][1]), "=v"(a[0][2]), "=v"(a[0][3]), "=v"(a[1][0]), "=v"(a[1][1]), "=v"(a[1][2]), "=v"(a[1][3]),
                 "=v"(a[2][0]), "=v"(a[2][1]), "=v"(a[2][2]), "=v"(a[2][3]), "=v"(a[3][0]), "=v"(a[3][1]), "=v"(a[3][2]), "=v"(a[3][3])
               : "n"(B)
               : G4W_ALL_AGPRS);
}
// The in-register epilogue of the 128 x 64 half H of the wave's tile = ONE virtual wave (wr, wc) of gemm256's 2 x 4 wave grid
// (gemm256_direct_epilogue.hpp, shared with gemm256.hip).  The accumulators are read from the AGPRs ROW BY ROW where stage 1 consumes
// them (16 registers at a time instead of the half's 128), with the folded-norm row scale applied on the way.
template <int EPI, int H, bool F8, bool MX, bool RG>
__device__ __forceinline__ void direct_epilogue_half(const GemmParams& p, int em0, int en0, int wr, int wc, int fr, int fq) {
  if constexpr (F8) {
    // W8A8: per-row activation scale x per-output-channel weight scale, the latter in the permuted-row order of the direct tile (gemm256.hip)
    // (MX: the activation's block scales were applied inside the MFMAs; a per-row scale is then optional — the folded RMSNorm's 1 / rms)
    float sa[8], sw[4][4];
#pragma unroll
    for (int m = 0; m < 8; ++m) sa[m] = (MX && !p.a_scale) ? 1.0f : p.a_scale[RG ? min(em0 + wr * 128 + m * 16 + fr, p.M - 1) : em0 + wr * 128 + m * 16 + fr];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      int sc;
      if (EPI == VSTAR_EPI_SILU_MUL) { const int jo = fq * 8 + (n >> 1) * 4; sc = wc * 64 + (jo >> 4) * 32 + (n & 1) * 16 + (jo & 15); }
      else if (EPI == VSTAR_EPI_NONE && p.rope_cs != nullptr && en0 < p.rope_cols) sc = (wc >> 1) * 128 + (n >> 1) * 64 + (wc & 1) * 32 + fq * 8 + (n & 1) * 4;
      else sc = wc * 64 + (n >> 1) * 32 + fq * 8 + (n & 1) * 4;
      const f32x4 t = *(const f32x4*)(p.w_scale + en0 + sc);
#pragma unroll
      for (int e = 0; e < 4; ++e) sw[n][e] = t[e];
    }
    gemm256_direct_epilogue<EPI, false, RG>(p, em0, en0, wr, wc, fr, fq, [&](auto mc, f32x4 (&a)[4]) {
      // (no contraction: gemm256 dequantises in a loop of its own, far from the bias add; here the two meet after inlining and
      // `acc * scale + bias` as ONE fma rounds differently — 4 of 524288 outputs in the first W8A8 bias test)
#pragma clang fp contract(off)
      constexpr int m = decltype(mc)::value;
      load_acc_row<H, m>(a);
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int e = 0; e < 4; ++e) a[n][e] *= sa[m] * sw[n][e];
    });
    return;
  }
  float rs_v[8];            // RMSNorm / LayerNorm folded into this linear: rstd[row] * (x . (W * norm_w)^T)
  if (p.row_scale) {
#pragma unroll
    for (int m = 0; m < 8; ++m) rs_v[m] = p.row_scale[RG ? min(em0 + wr * 128 + m * 16 + fr, p.M - 1) : em0 + wr * 128 + m * 16 + fr];
  }
  gemm256_direct_epilogue<EPI, true, RG>(p, em0, en0, wr, wc, fr, fq, [&](auto mc, f32x4 (&a)[4]) {
#pragma clang fp contract(off)      // (row scale x accumulator must not fuse with the bias add that follows after inlining, see the W8A8 branch)
    constexpr int m = decltype(mc)::value;
    load_acc_row<H, m>(a);
    if (p.row_scale) {
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int e = 0; e < 4; ++e) a[n][e] *= rs_v[m];
    }
  });
}

#ifndef VSTAR_LP_F16
// gate|up with a BLOCK-SCALED fp8 output (GemmParams::c_mx, mx.hpp): SiLU(gate) * up leaves as e4m3 bytes + one E8M0 byte per row and 32
// outputs — the down_proj input, quantised where it is produced.  The W rows of such a tile are DMA'd in an order of their own
// (piece_offsets) so that lane (fr, fq) ends up with SIXTEEN consecutive outputs of row group m — 8 from each 128 x 64 half of the
// wave's tile — i.e. one 16-byte store per row (4 lanes = 64 contiguous bytes; 8-byte stores measured -6.5 % on the whole GEMM), and a
// block of 32 = this lane's 16 values + lane ^ 16's.  Values are quantised from the 16-bit value the plain epilogue stores: same bytes as
// store + quantize_rows_mx (tests/test_mx_gpu.py).
__device__ __forceinline__ void mx_silu_epilogue(const GemmParams& p, int em0, int en0, int wr, int wc2, int fr, int fq) {
  float sa[8], sw[2][4][4];
#pragma unroll
  for (int m = 0; m < 8; ++m) sa[m] = p.a_scale ? p.a_scale[em0 + wr * 128 + m * 16 + fr] : 1.0f;
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const f32x4 t = *(const f32x4*)(p.w_scale + en0 + wc2 * 128 + fq * 32 + (n & 1) * 16 + h * 8 + (n >> 1) * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) sw[h][n][e] = t[e];
    }
  const int row0 = em0 + wr * 128 + fr;
  const int col0 = (en0 + wc2 * 128) / 2;
  uint8_t* qrow = (uint8_t*)p.C + (int64_t)row0 * p.ldc + col0 + fq * 16;
  uint32_t sc_lo = 0, sc_hi = 0;
  gemm_static_for<8>([&](auto mc) {
#pragma clang fp contract(off)      // the dequantisation of direct_epilogue_half, operation for operation
    constexpr int m = decltype(mc)::value;
    f32x4 a[2][4];
    load_acc_row<0, m>(a[0]);
    load_acc_row<1, m>(a[1]);
    float f[16], mx = 0.f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int e = 0; e < 4; ++e) a[h][n][e] *= sa[m] * sw[h][n][e];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        f[h * 8 + e] = rlp(act_silu_bf16(rlp(a[h][0][e])) * rlp(a[h][1][e]));
        f[h * 8 + 4 + e] = rlp(act_silu_bf16(rlp(a[h][2][e])) * rlp(a[h][3][e]));
      }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) mx = fmaxf(mx, fabsf(f[e]));
    mx = mx_max_row_pairs(mx);
    const uint32_t e8 = mx_e8m0(mx);
    const float inv = mx_inv_scale(e8);
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
    u32x4 o;
#pragma unroll
    for (int g = 0; g < 4; ++g) o[g] = mx_pack4(f[4 * g] * inv, f[4 * g + 1] * inv, f[4 * g + 2] * inv, f[4 * g + 3] * inv);
    __builtin_nontemporal_store(o, (u32x4*)qrow);
    qrow += 16 * p.ldc;
    if (m < 4) sc_lo |= e8 << (8 * (m & 3)); else sc_hi |= e8 << (8 * (m & 3));
  });
  if (!(fq & 1))      // rows row0 + 16 m, m = 0..7: eight consecutive bytes of the tile-major scale layout
    *(uint2*)(p.c_mx + mx_scale_offset(row0, (col0 >> 5) + (fq >> 1), p.M >> 7)) = make_uint2(sc_lo, sc_hi);
}
#endif

#ifndef VSTAR_LP_F16
// o_proj / down_proj with the residual stream leaving TWICE (GemmParams::c8): the 16-bit rows (C, + residual, as always) and their
// block-scaled fp8 copy — the A operand of the next q|k|v / gate|up, whose RMSNorm is folded (weight into W, 1 / rms as that GEMM's
// row scale, from the sum-of-squares partials written here).  W rows DMA'd so that lane (fr, fq) owns 32 CONSECUTIVE columns of row
// group m (fq * 32 + half * 16 + fragment * 4 + e): a block of 32 is one lane's — no cross-lane step — and the rows leave as four
// 16-byte stores (16-bit) + two (fp8).  Rounding points of the plain epilogue (f2lp(acc), then f2lp(that + residual)); the fp8 bytes are
// those of quantize_rows_mx over the stored rows (tests/test_mx_gpu.py).
__device__ __forceinline__ void mx_none_epilogue(const GemmParams& p, int em0, int en0, int wr, int wc2, int fr, int fq) {
  float sa[8], sw[2][4][4];
#pragma unroll
  for (int m = 0; m < 8; ++m) sa[m] = p.a_scale ? p.a_scale[em0 + wr * 128 + m * 16 + fr] : 1.0f;
  const int col = en0 + wc2 * 128 + fq * 32;
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const f32x4 t = *(const f32x4*)(p.w_scale + col + h * 16 + n * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) sw[h][n][e] = t[e];
    }
  const int row0 = em0 + wr * 128 + fr;
  lp_t* crow = (lp_t*)p.C + (int64_t)row0 * p.ldc + col;
  const lp_t* rrow = p.res ? p.res + (int64_t)row0 * p.ldr + col : nullptr;
  uint8_t* qrow = p.c8 + (int64_t)row0 * p.ldc8 + col;
  float* sq = p.sumsq_out ? p.sumsq_out + (int64_t)row0 * p.sumsq_ld + (col >> 6) : nullptr;
  uint32_t sc_lo = 0, sc_hi = 0;
  gemm_static_for<8>([&](auto mc) {
#pragma clang fp contract(off)
    constexpr int m = decltype(mc)::value;
    f32x4 a[2][4];
    load_acc_row<0, m>(a[0]);
    load_acc_row<1, m>(a[1]);
    lpx8 r[4];
    if (rrow) {
#pragma unroll
      for (int c = 0; c < 4; ++c) r[c] = *(const lpx8*)(rrow + c * 8);
      rrow += 16 * p.ldr;
    }
    float f[32], mx = 0.f, ss = 0.f;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int j = h * 16 + n * 4 + e;
          float v = rlp(a[h][n][e] * (sa[m] * sw[h][n][e]));
          if (rrow) v = rlp(v + lp2f((lp_t)r[j >> 3][j & 7]));
          f[j] = v;
        }
    lpx8 o[4];
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      o[j >> 3][j & 7] = (short)f2lp(f[j]);
      mx = fmaxf(mx, fabsf(f[j]));
      ss += f[j] * f[j];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) *(lpx8*)(crow + c * 8) = o[c];      // (cache-resident: the residual of the next epilogue)
    crow += 16 * p.ldc;
    if (sq) {
      const uint32_t u = __float_as_uint(ss);
      const auto w = __builtin_amdgcn_permlane16_swap(u, u, false, false);      // lane ^ 16: the other half of the 64-column span
      const float tot = __uint_as_float(w[0]) + __uint_as_float(w[1]);
      if (!(fq & 1)) sq[0] = tot;
      sq += 16 * p.sumsq_ld;
    }
    const uint32_t e8 = mx_e8m0(mx);
    const float inv = mx_inv_scale(e8);
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      u32x4 q;
#pragma unroll
      for (int g = 0; g < 4; ++g)
        q[g] = mx_pack4(f[c * 16 + 4 * g] * inv, f[c * 16 + 4 * g + 1] * inv, f[c * 16 + 4 * g + 2] * inv, f[c * 16 + 4 * g + 3] * inv);
      *(u32x4*)(qrow + c * 16) = q;
    }
    qrow += 16 * p.ldc8;
    if (m < 4) sc_lo |= e8 << (8 * (m & 3)); else sc_hi |= e8 << (8 * (m & 3));
  });
  *(uint2*)(p.c_mx + mx_scale_offset(row0, col >> 5, p.M >> 7)) = make_uint2(sc_lo, sc_hi);
}
#endif

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

#ifndef G4W_PF_LEAD
#define G4W_PF_LEAD 4      // the L2 prefetch of a K-tile runs this many K-tiles ahead of the tile being computed (its DMA: two ahead)
#endif

// PF: the loop text with the L2 prefetch duty (long K: the launcher decides).  F8: W8A8 (BASELINE config 5) — A and W are OCP fp8 e4m3
// bytes, a K-tile is still 128 bytes of every row (128 elements), one v_mfma_scale_f32_16x16x128_f8f6f4 replaces four bf16 MFMAs, the
// accumulators are dequantised (per-row x per-output-channel scale) on their way out of the AGPRs.
// MX (with F8): the A operand's E8M0 block scales (mx.hpp) ride along as a 17th DMA piece per K-tile and enter the MFMAs per lane.
// RG: the launch's last row tile is ragged (M % 256 != 0): A is fetched through a bounded descriptor (rows past M read as zero) and the
// epilogue skips those rows lane by lane; the interior-only instantiations keep the unguarded code (the guards cost 0.5 % of the step).
template <int EPI, bool PF, bool F8, bool MX, bool RG>
__global__ __launch_bounds__(256, 1) void gemm4w_kernel(const GemmParams p) {
  static_assert(!(MX && RG), "block-scaled operands: whole 256-row tiles only");
  static_assert(!MX || F8, "block scales exist for the fp8 operands only");
  constexpr int ES = F8 ? 1 : 2;                       // bytes per operand element
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc2 = wave & 1;
  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = p.N / BN;      // the last row tile may be ragged (its missing A rows read as zero)
  const int nwg = tiles_m * tiles_n;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const uint32_t ldsw = __builtin_amdgcn_readfirstlane(lds0 + wave * 8192);
  int cur_wkind = -1;
  uint32_t va[8], vw[8];
  int m0 = 0, n0 = 0, pi = 0;
  const char *abase = nullptr, *wbase = nullptr;
  const uint8_t* sbase = nullptr;                      // MX: the scale bytes of this tile's two row blocks, K-tile 0
  const uint32_t sstr = MX ? (uint32_t)(p.M >> 7) * 512u : 0u;      // ... and the distance to the next K-tile's
  // wave w moves the 8-row pieces g = 8 w + i of the A tile and of the W tile: per-lane byte offsets at k = 0, W rows in the order of
  // the tile's kind `cur_wkind` (1 plain / SiLU, 2 RoPE)
  auto piece_offsets = [&](int ln) {
    const int st_r = ln >> 3, st_c = ln & 7;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = (wave * 8 + i) * 8 + st_r;
      const int cg = st_c ^ ((row >> 1) & 7);
      va[i] = (uint32_t)((int64_t)row * p.lda * ES + cg * 16);
      const int wcr = row >> 6, n = (row >> 4) & 3, ii = row & 15;
      int wrow;
      if (EPI == VSTAR_EPI_SILU_MUL && F8 && p.c_mx) {      // block-scaled fp8 out: 16 consecutive outputs per lane (mx_silu_epilogue)
        wrow = (wcr >> 1) * 128 + (ii >> 2) * 32 + (n & 1) * 16 + (wcr & 1) * 8 + (n >> 1) * 4 + (ii & 3);
      } else if (EPI == VSTAR_EPI_NONE && F8 && p.c_mx) {   // ... 32 consecutive columns per lane (mx_none_epilogue)
        wrow = (wcr >> 1) * 128 + (ii >> 2) * 32 + (wcr & 1) * 16 + n * 4 + (ii & 3);
      } else if (EPI == VSTAR_EPI_SILU_MUL) {
        const int jo = (ii >> 2) * 8 + (n >> 1) * 4 + (ii & 3);
        wrow = wcr * 64 + (jo >> 4) * 32 + (n & 1) * 16 + (jo & 15);
      } else if (cur_wkind == 2) {
        wrow = (wcr >> 1) * 128 + (n >> 1) * 64 + (wcr & 1) * 32 + (ii >> 2) * 8 + (n & 1) * 4 + (ii & 3);
      } else {
        wrow = wcr * 64 + (n >> 1) * 32 + (ii >> 2) * 8 + (n & 1) * 4 + (ii & 3);
      }
      vw[i] = (uint32_t)((int64_t)wrow * p.K * ES + cg * 16);
    }
  };
  auto fresh_lane = []() {      // the lane id through an opaque instruction pair: ends every live range that derives from the old one
    int ln;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
    return ln;
  };
  // ---- tile id -> (m0, n0): XCD-aware bijective remap, then GROUP_M ordering (gemm256.hip); DMA source offsets of that tile ----
  auto set_tile = [&](int bid) {
    int t;
    {
      const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
      t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    constexpr int GROUP_M = 4;
    const int in_group = GROUP_M * tiles_n;
    const int grp = t / in_group;
    const int first_m = grp * GROUP_M;
    const int gsz = min(tiles_m - first_m, GROUP_M);
    const int rem = t - grp * in_group;
    pi = rem % gsz;
    m0 = (first_m + pi) * BM;
    n0 = (rem / gsz) * BN;
    bool rope_t = false;
    if constexpr (EPI == VSTAR_EPI_NONE) rope_t = p.rope_cs != nullptr && n0 < p.rope_cols;
    const int wkind = (EPI != VSTAR_EPI_SILU_MUL && rope_t) ? 2 : 1;
    // the per-lane part of a DMA source offset does not depend on the tile (the scalar bases carry it), only on the W row order of the
    // tile's kind.  (W8A8: recomputed around every loop statement anyway, see there.)
    if (wkind != cur_wkind) {
      cur_wkind = wkind;
      piece_offsets(lane);
    }
    abase = (const char*)p.A + (int64_t)m0 * p.lda * ES;
    wbase = (const char*)p.W + (int64_t)n0 * p.K * ES;
    if constexpr (MX) sbase = p.a_mx + (int64_t)(m0 >> 7) * 512;
  };
  // K-tiles 0 and 1 of the current tile -> LDS buffers 0 and 1 (32 DMA pieces per wave); the loop text starts behind a vmcnt(0)
  // bytes of A that exist from the tile's first row on: everything a full tile touches, or — ragged last row tile — up to the end of
  // row M - 1, so that the DMA of the missing rows is out of the descriptor's range and lands as ZEROS (never a read past the matrix)
  auto a_span = [&]() -> int {
    if constexpr (!RG) return -1;
    const int rows_left = p.M - m0;
    return rows_left >= BM ? -1 : (int)(((int64_t)rows_left - 1) * p.lda * ES + (int64_t)p.K * ES);
  };
  auto issue_head = [&]() {
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc((void*)abase, 0, a_span(), 0x00020000);
#endif
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if constexpr (!RG) {
          __builtin_amdgcn_global_load_lds((gptr_t)(abase + va[i] + t * 128), (lptr_t)(smem + t * 65536 + wave * 8192 + i * 1024), 16, 0, 0);
        } else {
#if defined(__HIP_DEVICE_COMPILE__)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(arsrc, (lptr_t)(smem + t * 65536 + wave * 8192 + i * 1024), 16, va[i], t * 128, 0, 0);
#endif
        }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i)
        __builtin_amdgcn_global_load_lds((gptr_t)(wbase + vw[i] + t * 128), (lptr_t)(smem + t * 65536 + 32768 + wave * 8192 + i * 1024), 16, 0, 0);
      if constexpr (MX)      // 1 KiB of scale bytes per K-tile: 256 B per wave, 4 B per lane
        __builtin_amdgcn_global_load_lds((gptr_t)(sbase + (int64_t)t * sstr + wave * 256 + (threadIdx.x & 63) * 4),
                                         (lptr_t)(smem + LDS_TOTAL + t * 1024 + wave * 256), 4, 0, 0);
    }
  };
  int bid = blockIdx.x;
  if (bid >= nwg) return;
  set_tile(bid);
  issue_head();
  int tl_tile = 0; (void)tl_tile;
  for (;;) {
    G4W_STAMP(0);
    // ---- fragment read addresses (buffer 0): row-major 128-B rows, chunk ^ ((row >> 1) & 7) ----
    uint32_t rd[4];
    {
      const int fr = lane & 15, fq = lane >> 4;
      const int swz = (fr >> 1) & 7;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int ch = ((kk * 4 + fq) ^ swz) * 16;
        rd[kk] = lds0 + (wr * 128 + fr) * 128 + ch;
        rd[2 + kk] = lds0 + 32768 + (wc2 * 128 + fr) * 128 + ch;
      }
    }
    // ---- L2 prefetch duty of this CU (tools/gen_gemm4w_asm.py::prefetch_ops): the 32 CUs of an XCD work on a patch of GROUP_M x 8
    // tiles; of the A row-tile's 256 lines per K-tile this CU touches the 32 of its patch column, of the W column-tile's 256 lines
    // the 64 of its patch row.  A wrong guess of the patch position (ragged groups, drifted workgroups) costs speed, never results.
    const int pj = ((bid >> 3) >> 2) & 7;
    const uint32_t kb_last = (uint32_t)(p.K * ES - 128);
    const uint32_t pf_lead = (uint32_t)G4W_PF_LEAD * 128 < kb_last ? (uint32_t)G4W_PF_LEAD * 128 : kb_last;
    const uint32_t pfa0 = (uint32_t)((int64_t)(pj * 32 + wave * 8 + (lane & 7)) * p.lda * ES);
    const uint32_t pfw0 = (uint32_t)((int64_t)((pi & 3) * 64 + wave * 16 + (lane & 15)) * p.K * ES);
    const uint32_t pfa = pfa0 + pf_lead, pfw = pfw0 + pf_lead, pfamax = pfa0 + kb_last, pfwmax = pfw0 + kb_last;
    uint32_t cnt = (uint32_t)(p.K * ES / 256 - 1);     // two K-tiles (of 128 bytes per row) per loop iteration, the last pair is peeled
    // buffer resource descriptors over the tile's rows (raw buffer: stride 0, no bound in practice, dword 3 = the gfx9 raw-buffer word)
    typedef __attribute__((ext_vector_type(4))) int i32x4;
    const i32x4 srda = {(int)(uint32_t)(uintptr_t)abase, (int)(((uintptr_t)abase >> 32) & 0xffff), a_span(), 0x00020000};
    const i32x4 srdw = {(int)(uint32_t)(uintptr_t)wbase, (int)(((uintptr_t)wbase >> 32) & 0xffff), -1, 0x00020000};
    uint32_t koff = 256;                               // K-tiles 0 and 1 are in flight (issue_head): the loop's first pieces are K-tile 2
#define G4W_OPERANDS(CLOB)                                                                                                           \
    : [cnt] "+s"(cnt), [koff] "+s"(koff)                                                                                             \
    : [srda] "s"(srda), [srdw] "s"(srdw), [ldsw] "s"(ldsw), [rd0] "v"(rd[0]), [rd1] "v"(rd[1]), [rd2] "v"(rd[2]),                    \
      [rd3] "v"(rd[3]), [va0] "v"(va[0]), [va1] "v"(va[1]), [va2] "v"(va[2]), [va3] "v"(va[3]),                                    \
      [va4] "v"(va[4]), [va5] "v"(va[5]), [va6] "v"(va[6]), [va7] "v"(va[7]), [vw0] "v"(vw[0]),                                    \
      [vw1] "v"(vw[1]), [vw2] "v"(vw[2]), [vw3] "v"(vw[3]), [vw4] "v"(vw[4]), [vw5] "v"(vw[5]),                                    \
      [vw6] "v"(vw[6]), [vw7] "v"(vw[7]), [pfa] "v"(pfa), [pfw] "v"(pfw), [pfamax] "v"(pfamax), [pfwmax] "v"(pfwmax)               \
    : CLOB
    if constexpr (F8) {
      // The W8A8 text uses 223 VGPRs (two A fragment sets + the rolling W set = 192): its per-lane operands are PINNED to the
      // registers the text reads them from, and everything per-lane is recomputed from a fresh lane id right here and again behind the
      // statement — a value that lives across it (or an input that needs a register of its own) has nowhere to be.
      typedef __attribute__((ext_vector_type(8))) int i32x8;
      piece_offsets(fresh_lane());
      i32x8 va8 = {(int)va[0], (int)va[1], (int)va[2], (int)va[3], (int)va[4], (int)va[5], (int)va[6], (int)va[7]};
      i32x8 vw8 = {(int)vw[0], (int)vw[1], (int)vw[2], (int)vw[3], (int)vw[4], (int)vw[5], (int)vw[6], (int)vw[7]};
      i32x4 rd4 = {(int)rd[0], (int)rd[1], (int)rd[2], (int)rd[3]};
      i32x4 pf4 = {(int)pfa, (int)pfw, (int)pfamax, (int)pfwmax};
#define G4W_OPERANDS_F8                                                                                                              \
    : [cnt] "+s"(cnt), [koff] "+s"(koff), [va] "+{v[200:207]}"(va8), [vw] "+{v[208:215]}"(vw8), [pf] "+{v[218:221]}"(pf4)            \
    : [srda] "s"(srda), [srdw] "s"(srdw), [ldsw] "s"(ldsw), [rd] "{v[192:195]}"(rd4)                                                \
    : GEMM4W_CLOBBERS_F8
      if constexpr (MX) {
        const i32x4 srds = {(int)(uint32_t)(uintptr_t)sbase, (int)(((uintptr_t)sbase >> 32) & 0xffff), -1, 0x00020000};
        uint32_t soff = 2 * sstr;                        // K-tiles 0 and 1 are in flight
        const uint32_t ldss = __builtin_amdgcn_readfirstlane(lds0 + LDS_TOTAL + wave * 256);
        const int ln = fresh_lane();
        int sr = (int)(lds0 + LDS_TOTAL + wr * 512 + ln * 8), so = wave * 256 + ln * 4;
#define G4W_OPERANDS_MX                                                                                                              \
    : [cnt] "+s"(cnt), [koff] "+s"(koff), [soff] "+s"(soff), [va] "+{v[200:207]}"(va8), [vw] "+{v[208:215]}"(vw8),                    \
      [pf] "+{v[218:221]}"(pf4)                                                                                                      \
    : [srda] "s"(srda), [srdw] "s"(srdw), [srds] "s"(srds), [ldsw] "s"(ldsw), [ldss] "s"(ldss), [sstr] "s"(sstr),                   \
      [rd] "{v[192:195]}"(rd4), [sr] "{v228}"(sr), [so] "{v229}"(so)                                                                \
    : GEMM4W_CLOBBERS_MX
        if constexpr (PF) asm volatile(GEMM4W_LOOP_ASM_MX_PF G4W_OPERANDS_MX);
        else asm volatile(GEMM4W_LOOP_ASM_MX G4W_OPERANDS_MX);
#undef G4W_OPERANDS_MX
      } else {
        if constexpr (PF) asm volatile(GEMM4W_LOOP_ASM_F8_PF G4W_OPERANDS_F8);
        else asm volatile(GEMM4W_LOOP_ASM_F8 G4W_OPERANDS_F8);
      }
#undef G4W_OPERANDS_F8
      cur_wkind = -1;             // the offsets died with the statement: the next set_tile recomputes them (from the fresh lane id below)
      lane = fresh_lane();
    } else {
      if constexpr (PF) asm volatile(GEMM4W_LOOP_ASM_PF G4W_OPERANDS(GEMM4W_CLOBBERS));
      else asm volatile(GEMM4W_LOOP_ASM G4W_OPERANDS(GEMM4W_CLOBBERS));
    }
#undef G4W_OPERANDS
    G4W_STAMP(1);
    // ---- next tile: its K-tiles 0 and 1 go into the (dead: the loop text ends behind a barrier) LDS buffers NOW, so that the
    // pipeline fill overlaps this tile's epilogue; this tile's coordinates stay in em0 / en0 ----
    const int em0 = m0, en0 = n0;
    bid += gridDim.x;
    const bool has_next = bid < nwg;
    if (has_next) {
      set_tile(bid);
      issue_head();
    }
    G4W_STAMP(2);
    // ---- epilogue: the wave's 128 x 128 as the two virtual waves (wr, 2 wc2) and (wr, 2 wc2 + 1) of gemm256's grid ----
#ifdef G4W_ABL_NOEPI      // ablation build (timing only): what the whole epilogue costs
    if (false) {
#else
    if (!(p.debug_flags & 2)) {
#endif
      const int fr = lane & 15, fq = lane >> 4;
      bool done = false;
#ifndef VSTAR_LP_F16
      if constexpr (F8 && EPI == VSTAR_EPI_SILU_MUL) {
        if (p.c_mx) {
          mx_silu_epilogue(p, em0, en0, wr, wc2, fr, fq);
          done = true;
        }
      }
      if constexpr (F8 && EPI == VSTAR_EPI_NONE) {
        if (p.c_mx) {
          mx_none_epilogue(p, em0, en0, wr, wc2, fr, fq);
          done = true;
        }
      }
#endif
      if (!done) {
        direct_epilogue_half<EPI, 0, F8, MX, RG>(p, em0, en0, wr, wc2 * 2, fr, fq);
        direct_epilogue_half<EPI, 1, F8, MX, RG>(p, em0, en0, wr, wc2 * 2 + 1, fr, fq);
      }
    }
    G4W_STAMP(3);
#ifdef G4W_TIMELINE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (the loop text starts with the same wait: here it gets its own stamp)
    G4W_STAMP(4);
    ++tl_tile;
#endif
    if (!has_next) break;
  }
}

template <int EPI, bool PF, bool F8, bool MX, bool RG>
hipError_t launch_rg(const GemmParams& p, hipStream_t s) {
  if (gemm_plan_only()) return hipSuccess;
  static bool attr_done = false;
  auto kern = gemm4w_kernel<EPI, PF, F8, MX, RG>;
  constexpr int LDS_BYTES = LDS_TOTAL + (MX ? 2048 : 0);
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  const int n_cu = gemm_device_cus();
  const int tiles = ((p.M + BM - 1) / BM) * (p.N / BN);
  hipLaunchKernelGGL(kern, dim3(tiles < n_cu ? tiles : n_cu), dim3(256), LDS_BYTES, s, p);
  return hipGetLastError();
}
template <int EPI, bool PF, bool F8 = false, bool MX = false>
hipError_t launch(const GemmParams& p, hipStream_t s) {
  if constexpr (!MX) {      // ragged launches: the plain loop text (with the L2 prefetch duty down_proj measured 1384 instead of 1441 TFLOP/s)
    if (p.M % BM) return launch_rg<EPI, false, F8, MX, true>(p, s);
  }
  return launch_rg<EPI, PF, F8, MX, false>(p, s);
}

}  // namespace

#if defined(G4W_TIMELINE) && !defined(VSTAR_LP_F16)
extern "C" int vstar_debug_gemm4w_timeline(unsigned long long* out) {     // 2 x 64 x 8 stamps
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g4w_timeline), sizeof(g4w_timeline));
}
#endif

// Launches made of interior tiles that take gemm256's direct epilogue (see the header comment for the domain).
bool gemm4w_eligible(const GemmParams& p, int epilogue, bool out_f32) {
#ifdef VSTAR_LP_F16
  if (p.a_scale || p.a_mx || p.c_mx) return false;
#endif
  if (out_f32 || p.norm_w) return false;
  if (p.a_mx && ((epilogue != VSTAR_EPI_NONE && epilogue != VSTAR_EPI_SILU_MUL) || ((uintptr_t)p.a_mx & 3))) return false;      // block-scaled A
  if (p.c_mx) {      // block-scaled output: SiLU(gate) * up as fp8 only (C), or the residual stream as 16-bit rows (C) + fp8 copy (c8)
    if (!(p.a_scale || p.a_mx) || ((uintptr_t)p.c_mx & 7)) return false;
    if (epilogue == VSTAR_EPI_SILU_MUL) { if (p.ldc % 16 || p.c8) return false; }
    else if (epilogue == VSTAR_EPI_NONE) { if (!p.c8 || ((uintptr_t)p.c8 & 15) || p.ldc8 % 16 || p.bias || p.rope_cs || p.stats_sum) return false; }
    else return false;
  } else if (p.c8) return false;
  if (p.a_scale || p.a_mx) {      // W8A8: the two epilogues the LLaMA linears use; K counts fp8 elements, two K-tiles of 128 per loop iteration
    if (!p.w_scale || (epilogue != VSTAR_EPI_NONE && epilogue != VSTAR_EPI_SILU_MUL) || p.K % 256 || p.row_scale || (p.sumsq_out && !p.c8)) return false;
    static const bool f8_on = [] { const char* e = getenv("VSTAR_GEMM4W_F8"); return !e || atoi(e) != 0; }();
    if (!f8_on) return false;
  }
  if (epilogue != VSTAR_EPI_NONE && epilogue != VSTAR_EPI_QUICK_GELU && epilogue != VSTAR_EPI_RELU && epilogue != VSTAR_EPI_SILU_MUL) return false;
  // a ragged last row tile is fine (A rows past M read as zero, the epilogue skips them) — except for the block-scaled operands, whose
  // scale layout is per 128-row block, and for K < 4096 (the ViT shapes: gemm256 is the better kernel for them)
  if (p.M % BM && (p.a_mx || p.c_mx || p.K < 4096)) return false;
  if (p.M < 1024 || p.N % BN || p.K % 128 || p.K < 128) return false;
  if (p.a_group > 0 || p.c_group > 0 || (p.debug_flags & 5)) return false;
  if (((uintptr_t)p.C & 15) || (p.ldc % 8)) return false;
  if (p.res && (epilogue == VSTAR_EPI_SILU_MUL || ((uintptr_t)p.res & 15) || (p.ldr % 8))) return false;
  if (p.bias && ((uintptr_t)p.bias & 15)) return false;
  if (p.sumsq_out && epilogue != VSTAR_EPI_NONE) return false;
  if ((int64_t)255 * p.lda * 2 + (int64_t)p.K * 2 >= (1ll << 31) || (int64_t)256 * p.K * 2 >= (1ll << 31)) return false;
  return true;
}

bool gemm_mx_supported(const GemmParams& p, int epilogue) { return gemm4w_eligible(p, epilogue, false); }

hipError_t gemm4w_lp(const GemmParams& p, int epilogue, hipStream_t s) {
  // L2 prefetch duty: measured on the MI355X (profiles/r06_gemm4w_ab.txt) it gains 4 - 15 % with the clock unconstrained (zero
  // operands) but, on real operands under the 1400 W cap, only where the stalls it removes are long — K = 11008 (down_proj: an A
  // operand of 451 MB, past the Infinity Cache) +2 - 5 %, K = 4096 -1 - 2 % (its extra requests cost more energy than the shorter
  // stalls return).  VSTAR_GEMM4W_PF = 0 / 1 forces it off / on (A/B runs).
  static const int env_pf = [] { const char* e = getenv("VSTAR_GEMM4W_PF"); return e ? atoi(e) : -1; }();
  // fp8: the same picture at 64 crops — K = 11008 +3 % (2828 -> 2911 TFLOP/s), K = 4096 within +-1 %
  const bool pf = env_pf >= 0 ? env_pf != 0 : (p.a_scale || p.a_mx) ? p.K >= 8192 : (int64_t)p.K * 2 >= 16384;
#ifndef VSTAR_LP_F16
  if (p.a_mx) {
    if (epilogue == VSTAR_EPI_NONE) return pf ? launch<VSTAR_EPI_NONE, true, true, true>(p, s) : launch<VSTAR_EPI_NONE, false, true, true>(p, s);
    if (epilogue == VSTAR_EPI_SILU_MUL) return pf ? launch<VSTAR_EPI_SILU_MUL, true, true, true>(p, s) : launch<VSTAR_EPI_SILU_MUL, false, true, true>(p, s);
    return hipErrorInvalidValue;
  }
  if (p.a_scale) {
    if (epilogue == VSTAR_EPI_NONE) return pf ? launch<VSTAR_EPI_NONE, true, true>(p, s) : launch<VSTAR_EPI_NONE, false, true>(p, s);
    if (epilogue == VSTAR_EPI_SILU_MUL) return pf ? launch<VSTAR_EPI_SILU_MUL, true, true>(p, s) : launch<VSTAR_EPI_SILU_MUL, false, true>(p, s);
    return hipErrorInvalidValue;
  }
#endif
#define G4W_CASE(E) case E: return pf ? launch<E, true>(p, s) : launch<E, false>(p, s);
  switch (epilogue) {
    G4W_CASE(VSTAR_EPI_NONE)
    G4W_CASE(VSTAR_EPI_QUICK_GELU)
    G4W_CASE(VSTAR_EPI_RELU)
    G4W_CASE(VSTAR_EPI_SILU_MUL)
  }
#undef G4W_CASE
  return hipErrorInvalidValue;
}

}  // namespace VS_NS
