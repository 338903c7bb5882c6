// gemm256.hip — the large-shape bf16 / fp16 / W8A8 MFMA GEMM for gfx950: 256x256x64 block tile, 8 waves (2 M x 4 N, 128x64 each),
// 128 KiB LDS ring (2 K-tiles), global_load_lds DMA running 6 quarter-tiles ahead behind COUNTED vmcnt waits (the first
// K-tile of the next output tile is already in flight during the epilogue), and two
// wave groups staggered by one barrier so that on every SIMD one wave is in its MFMA cluster while its partner issues
// ds_reads / DMA ("8-phase" structure of the CDNA4 guide, §5 "256^2 8-phase template", re-derived for this layout).
//
// Same math / epilogues / row maps as gemm.hip (which stays the kernel for small M, small N or odd K/64).
//
// Schedule.  A K-tile is consumed in 2 phases, each = L (LDS fragment reads + two DMA pieces) | barrier | M (32 MFMA = two
// 64x32 quadrants of the wave's tile over K=64) | barrier:
//      phase A: read A[m-half 0] (8 x b128) + W[n-half 0] (4) + W[n-half 1] (4)  -> quadrants (0,0),(0,1)
//      phase B: read A[m-half 1] (8)                                              -> quadrants (1,1),(1,0)
// (four phases of 16 MFMA were 4 % slower: a barrier interval costs ~100 cycles of pure synchronisation.)
// The K-tile is DMA'd in 4 pieces (a0, w0, w1, a1; 16 KiB = 2 x 1-KiB DMA per wave each), six pieces ahead: phase A(T)
// issues pieces w1,a1 of tile T+1, phase B(T) issues a0,w0 of tile T+2; each overwrites an LDS region whose last reader
// (of either group) retired (lgkmcnt(0)) before an earlier barrier, and each wave then waits with a COUNTED vmcnt until
// everything the next phase reads has landed (vmcnt(8) in A, vmcnt(6) in B) before the barrier that precedes that phase.
// Group 1 (waves 4-7) runs one barrier behind group 0, so L of one group always overlaps M of the other.
// (Six pieces is the measured optimum: four cost 14 - 24 %, eight — a ten-slot piece ring over all of LDS — 4 - 5 %; round 5,
// tools/experiments/gemm256_ring10/.)
//
// Epilogue (round 5).  Interior tiles of a launch with aligned operands finish IN THE ACCUMULATOR REGISTERS: the wave's 64 W
// rows are DMA'd in a permuted order so that a lane's four fragments are two runs of 8 consecutive output columns — see
// `dir_launch` below and gemm256_direct_epilogue.hpp (the code, shared with gemm4w.hip since round 6).  Edge tiles, row-mapped outputs, fp32 outputs and the erf GELU keep the LDS-transposed epilogue.  No
// per-lane value lives across the K loop (the lane id is re-read after it), so no instantiation spills.
#include "common.hpp"
#include "kernels.hpp"
#include "gemm_epilogue.hpp"
#include "gemm256_direct_epilogue.hpp"
#include <type_traits>

namespace VS_NS {

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int A_BYTES = BM * BK * 2;               // 32 KiB
constexpr int TILE_BYTES = 2 * A_BYTES;            // A + W = 64 KiB per K-tile
constexpr int LDS_TOTAL = 2 * TILE_BYTES;          // 128 KiB ring

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

#define BAR()                                          \
  do {                                                 \
    __builtin_amdgcn_sched_barrier(0);                 \
    asm volatile("s_barrier" ::: "memory");            \
    __builtin_amdgcn_sched_barrier(0);                 \
  } while (0)
#define LGKM0_BAR()                                                 \
  do {                                                              \
    __builtin_amdgcn_sched_barrier(0);                              \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); \
    __builtin_amdgcn_sched_barrier(0);                              \
  } while (0)

#ifdef GEMM_TIMELINE   // diagnostic builds only (tools/gemm_timeline.py): wall-clock stamps (100 MHz) of the tile phases
__device__ unsigned long long g_timeline[2][2][64][8];      // [block 0 | block 100][wave 0 | wave 4][tile][point]
#define TL_STAMP(pt)                                                                                              \
  do {                                                                                                            \
    if (lane == 0 && (wave & 3) == 0 && (blockIdx.x == 0 || blockIdx.x == 100) && tl_tile < 64)                   \
      g_timeline[blockIdx.x ? 1 : 0][wave >> 2][tl_tile][pt] = wall_clock64();                                    \
  } while (0)
#else
#define TL_STAMP(pt) do { } while (0)
#endif

template <int EPI, bool OUT_F32, bool F8>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(const GemmParams p) {
  constexpr int ES = F8 ? 1 : 2;                    // bytes per operand element; a K-tile is 128 bytes of every row either way
  constexpr int BKE = 128 / ES;                     // elements per K-tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  int lane = tid & 63;                             // (re-read after every K loop, see there)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int group = wave >> 2;                     // 0: waves 0-3, 1: waves 4-7 (one of each per SIMD)

  // ---- tile id: XCD-aware bijective remap, then GROUP_M ordering ----
  const int tiles_m = (p.M + BM - 1) / BM;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int nwg = tiles_m * tiles_n;
  // persistent: this workgroup walks tiles blockIdx.x, +gridDim.x, ... (gridDim.x = #CUs, a multiple of 8, so a
  // workgroup's tiles keep its XCD in the remap below).  The previous tile's output stores drain while the next
  // tile's first DMA pieces are in flight, and there is no per-tile workgroup launch.
  // The first K-tile of the NEXT output tile is DMA'd (into ring buffer 0, dead by then) BEFORE the epilogue of the
  // current one, whose LDS slabs live in ring buffer 1 (where the last K-tile — K/64 is even — was just consumed): the
  // pipeline fill of a tile overlaps the output stores of its predecessor.
  int st_r = lane >> 3, st_c = lane & 7;
  uint32_t src_off[4][2];   // BYTE offset of this lane's 16-B chunk at k0 = 0, per piece type and u
  int lds_off[4][2];        // wave-uniform LDS byte offset of the 1-KiB piece inside a K-tile buffer
  int m0 = 0, n0 = 0;
  uint32_t a_base = 0, w_base = 0;     // this tile's scalar byte offsets into A and W
  int cur_wkind = -1;                  // what src_off currently holds (see set_tile)
  bool cur_a_general = true;
  // ---- direct (in-register) epilogue, round 5 ----
  // The MFMA leaves lane (fr, fq) with output columns n*16 + fq*4 + e of fragment n: 8-byte pieces, which is why the bf16 epilogue
  // used to transpose through LDS.  Which W ROW a given LDS row holds is free, though (the DMA source address is per lane): on
  // interior tiles of a launch with aligned operands the wave's 64 W rows are DMA'd in the order
  //      LDS row (n, i)  <-  W row (n>>1)*32 + (i>>2)*8 + (n&1)*4 + (i&3)
  // so that the lane's fragments 0|1 are 8 CONSECUTIVE output columns (16 bytes) at fq*8 and fragments 2|3 the 8 at 32 + fq*8:
  // bias / activation / residual / RoPE / statistics and two 16-byte stores per row happen in the accumulator registers — no LDS
  // slab, no transpose, no rolled row loop.  RoPE tiles (two heads of 128 columns per tile, wave pair = head) instead take
  // d = (wc&1)*32 + fq*8.. and d + 64 into ONE lane (fragments 0|1 and 2|3), the rotate-half partner without any exchange; the
  // SiLU tiles (storage rows: 16 gate rows, 16 up rows, ...) put gate / up of 8 consecutive outputs into fragments 0,2 / 1,3.
  // Same dot products, same k order, same rounding points, same statistics tree: bit-identical to the LDS epilogue.
  // (not for the erf GELU: 128 inlined erff bodies per lane spill; its two launches per step keep the rolled LDS epilogue)
  const bool dir_launch = !OUT_F32 && EPI != VSTAR_EPI_GELU && !(p.debug_flags & 7) && p.c_group <= 0 && (((uintptr_t)p.C & 15) == 0) && (p.ldc % 8 == 0) &&
                          (p.res == nullptr || (EPI != VSTAR_EPI_SILU_MUL && (((uintptr_t)p.res & 15) == 0) && (p.ldr % 8 == 0))) &&
                          (p.bias == nullptr || (((uintptr_t)p.bias & 15) == 0));
  bool dir_tile = false;
  // ---- tile id -> (m0, n0): XCD-aware bijective remap, then GROUP_M ordering; DMA source offsets of that tile ----
  // DMA pieces: piece type j in {a0, w0, w1, a1}; each wave moves row-groups g = 2*wave + u (u = 0,1)
  //   A piece (m-half mh): 8-row group g -> tile rows (g>>3)*128 + mh*64 + (g&7)*8 ..+8
  //   W piece (n-half nh): 8-row group g -> tile rows (g>>2)*64  + nh*32 + (g&3)*8 ..+8
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
    m0 = (first_m + rem % gsz) * BM;
    n0 = (rem / gsz) * BN;
    dir_tile = dir_launch && m0 + BM <= p.M && n0 + BN <= p.N;
    bool rope_t = false;
    if constexpr (EPI == VSTAR_EPI_NONE) rope_t = p.rope_cs != nullptr && n0 < p.rope_cols;
    // The per-lane part of a DMA source offset does not depend on the tile: W rows are never clamped (W is padded) and an interior
    // A tile under the identity row map is m0 * lda away from tile 0's.  So the tile enters through two SCALAR bases (a_base,
    // w_base, added to the scalar pointer in issue_piece) and the eight per-lane offsets are recomputed only when their KIND changes
    // (W row order: identity / direct / RoPE; A: interior vs clamped-or-mapped rows) — a few tiles per launch instead of every
    // tile's ~250 VALU instructions on the critical path between two K loops (1.3 - 1.7 us per tile, tools/gemm_timeline.py).
    const int wkind = !dir_tile ? 0 : ((EPI != VSTAR_EPI_SILU_MUL && rope_t) ? 2 : 1);
    const bool a_general = (m0 + BM > p.M) || p.a_group > 0;
    w_base = (uint32_t)((int64_t)n0 * p.K * ES);
    a_base = a_general ? 0u : (uint32_t)((int64_t)m0 * p.lda * ES);
    const bool redo_w = wkind != cur_wkind, redo_a = a_general || cur_a_general;
    cur_wkind = wkind;
    cur_a_general = a_general;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int g = 2 * wave + u;
        const bool isA = (j == 0 || j == 3);
        const int half = (j == 0 || j == 1) ? 0 : 1;     // a0,w0 -> half 0 ; w1,a1 -> half 1
        const int row0 = isA ? ((g >> 3) * 128 + half * 64 + (g & 7) * 8) : ((g >> 2) * 64 + half * 32 + (g & 3) * 8);
        const int row = row0 + st_r;
        const int cg = st_c ^ ((row >> 1) & 7);
        lds_off[j][u] = (isA ? 0 : A_BYTES) + row0 * 128;
        if (isA) {
          if (redo_a) {
            if (a_general) {
              int ar = m0 + row;
              ar = ar < p.M ? ar : p.M - 1;
              src_off[j][u] = (uint32_t)(gemm_map_row(ar, p.a_group, p.a_gstride, p.a_off) * p.lda * ES + cg * 16);
            } else {
              src_off[j][u] = (uint32_t)((int64_t)row * p.lda * ES + cg * 16);
            }
          }
        } else if (redo_w) {
          int wrow = row;
          if (wkind != 0) {
            const int wcr = row >> 6, n = (row >> 4) & 3, i = row & 15;
            if (EPI == VSTAR_EPI_SILU_MUL) {
              const int jo = (i >> 2) * 8 + (n >> 1) * 4 + (i & 3);              // output column inside the wave's 32
              wrow = wcr * 64 + (jo >> 4) * 32 + (n & 1) * 16 + (jo & 15);      // its gate (n even) / up (n odd) storage row
            } else if (wkind == 2) {
              wrow = (wcr >> 1) * 128 + (n >> 1) * 64 + (wcr & 1) * 32 + (i >> 2) * 8 + (n & 1) * 4 + (i & 3);
            } else {
              wrow = wcr * 64 + (n >> 1) * 32 + (i >> 2) * 8 + (n & 1) * 4 + (i & 3);
            }
          }
          src_off[j][u] = (uint32_t)((int64_t)wrow * p.K * ES + cg * 16);
        }
      }
    }
  };
  // piece type J is a compile-time constant at every call site, so src_off / lds_off stay in registers
  auto issue_piece = [&](auto jc, int T) {
    constexpr int J = decltype(jc)::value;
    char* base = smem + (T & 1) * TILE_BYTES;
#ifdef GEMM_ABL_NODMA          // ablation builds (tools/power_probe.py --ablations): results are garbage, timing / power are the point
    return;
#endif
#ifdef GEMM_ABL_NOKADV
    const char* gb = ((J == 0 || J == 3) ? (const char*)p.A + a_base : (const char*)p.W + w_base) + (T & 1) * 128;
#else
    const char* gb = ((J == 0 || J == 3) ? (const char*)p.A + a_base : (const char*)p.W + w_base) + T * 128;
#endif
#pragma unroll
    for (int u = 0; u < 2; ++u)
      __builtin_amdgcn_global_load_lds((gptr_t)(gb + src_off[J][u]), (lptr_t)(base + lds_off[J][u]), 16, 0, 0);
  };

  // ---- fragment read offsets (same row-major + (row>>1)&7 chunk swizzle as gemm.hip) ----
  const int wr = wave >> 2, wc = wave & 3;
  int fr = lane & 15, fq = lane >> 4;
  const int swz = (fr >> 1) & 7;
  int a_rd[2], w_rd[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int ch = ((kk * 4 + fq) ^ swz) * 16;
    a_rd[kk] = (wr * 128 + fr) * 128 + ch;
    w_rd[kk] = A_BYTES + (wc * 64 + fr) * 128 + ch;
  }

  int bid = blockIdx.x;
  if (bid >= nwg) return;
  // Experiment switch (VSTAR_GEMM_XCD_STAGGER=n -> debug bits 8..15): XCD x (= blockIdx % 8) starts x * n * ~0.26 us late, so that
  // the eight XCDs — which share no L2 and therefore do not pull each other back into step — reach their store bursts at
  // different times (all 256 CUs storing 128 KB at once is a 32-MiB burst against ~5 TB/s of write bandwidth).
  if (const int stg = (p.debug_flags >> 8) & 0xff) {
    for (int i = 0; i < (int)(blockIdx.x & 7) * stg; ++i) __builtin_amdgcn_s_sleep(8);
  }
  set_tile(bid);
  issue_piece(std::integral_constant<int, 0>{}, 0);
  issue_piece(std::integral_constant<int, 1>{}, 0);
  issue_piece(std::integral_constant<int, 2>{}, 0);
  issue_piece(std::integral_constant<int, 3>{}, 0);
  int tl_tile = 0; (void)tl_tile;
  for (;;) {
  TL_STAMP(0);
  f32x4 acc[8][4];
#pragma unroll
  for (int m = 0; m < 8; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- prologue: K-tile 0 (pieces a0, w0, w1, a1) was issued before the previous tile's epilogue (or at kernel start);
  // add a0, w0 of K-tile 1 once every wave has left its epilogue slab (ring buffer 1), then wait until a0, w0, w1 of
  // K-tile 0 have landed.  Issue order: [4 pieces of K-tile 0][the previous epilogue's stores/loads][these 2 pieces]; vmcnt
  // retires in order, so at most 6 outstanding leaves only the 2 new pieces (4 DMAs) + 2 older ops in flight.  Group 1 runs
  // one barrier behind group 0.
  BAR();
  issue_piece(std::integral_constant<int, 0>{}, 1);
  issue_piece(std::integral_constant<int, 1>{}, 1);
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  BAR();
  if (group == 1) BAR();
  TL_STAMP(1);

  // Two phases per K-tile (4 barrier rendezvous instead of 8; a barrier interval costs ~100 cycles of pure sync):
  //   phase A: read A[m-half 0], W[n-half 0], W[n-half 1] (16 x b128) -> quadrants (0,0),(0,1)   = 32 MFMA
  //   phase B: read A[m-half 1] (8 x b128)                            -> quadrants (1,1),(1,0)   = 32 MFMA
  // DMA: phase A(T) issues pieces 4T+6, 4T+7 (w1, a1 of tile T+1), phase B(T) issues 4T+8, 4T+9 (a0, w0 of tile T+2) —
  // each overwrites a region whose last reader retired at least one phase (= one barrier on both groups) earlier — and
  // waits until everything the NEXT phase reads has landed: vmcnt(8) in A (4 pieces may stay in flight), vmcnt(6) in B.
  lpx8 af[8], w0f[4], w1f[4];
  const int nkt = p.K / BKE;
  for (int T = 0; T < nkt; ++T) {
    const char* base = smem + (T & 1) * TILE_BYTES;
#define MFMA_PAIR(mh, WFA, nha, WFB, nhb)                                                \
  {                                                                                      \
    __builtin_amdgcn_s_setprio(1);                                                       \
    if constexpr (F8) {                                                                  \
      _Pragma("unroll") for (int m = 0; m < 4; ++m)                                      \
        _Pragma("unroll") for (int n = 0; n < 2; ++n) {                                  \
          acc[(mh) * 4 + m][(nha) * 2 + n] = mfma_16x16x128_fp8(                         \
              WFA[n], WFA[2 + n], af[m], af[4 + m], acc[(mh) * 4 + m][(nha) * 2 + n]);   \
          acc[(mh) * 4 + m][(nhb) * 2 + n] = mfma_16x16x128_fp8(                         \
              WFB[n], WFB[2 + n], af[m], af[4 + m], acc[(mh) * 4 + m][(nhb) * 2 + n]);   \
        }                                                                                \
    } else {                                                                             \
      _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                   \
        _Pragma("unroll") for (int m = 0; m < 4; ++m)                                    \
          _Pragma("unroll") for (int n = 0; n < 2; ++n) {                                \
            acc[(mh) * 4 + m][(nha) * 2 + n] = mfma_16x16x32(                            \
                WFA[kk * 2 + n], af[kk * 4 + m], acc[(mh) * 4 + m][(nha) * 2 + n]);      \
            acc[(mh) * 4 + m][(nhb) * 2 + n] = mfma_16x16x32(                            \
                WFB[kk * 2 + n], af[kk * 4 + m], acc[(mh) * 4 + m][(nhb) * 2 + n]);      \
          }                                                                              \
    }                                                                                    \
    __builtin_amdgcn_s_setprio(0);                                                       \
    BAR();                                                                               \
  }
    // ---- phase A ----
#ifdef GEMM_ABL_NOREAD
    if (T == 0) {
#endif
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        w0f[kk * 2 + n] = *(const lpx8*)(base + w_rd[kk] + n * 2048);
        w1f[kk * 2 + n] = *(const lpx8*)(base + w_rd[kk] + (2 + n) * 2048);
      }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int m = 0; m < 4; ++m) af[kk * 4 + m] = *(const lpx8*)(base + a_rd[kk] + m * 2048);
#ifdef GEMM_ABL_NOREAD
    }
#endif
    if (T + 1 < nkt) {
      issue_piece(std::integral_constant<int, 2>{}, T + 1);
      issue_piece(std::integral_constant<int, 3>{}, T + 1);
#ifdef GEMM_SHALLOW      // sensitivity probe (variant build): two pieces less in flight at every wait = a 4-deep instead of a 6-deep DMA pipeline
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
#else
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#endif
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    LGKM0_BAR();
    MFMA_PAIR(0, w0f, 0, w1f, 1)
    // ---- phase B ----
#ifndef GEMM_ABL_NOREAD
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int m = 0; m < 4; ++m) af[kk * 4 + m] = *(const lpx8*)(base + a_rd[kk] + (4 + m) * 2048);
#endif
    if (T + 2 < nkt) {
      issue_piece(std::integral_constant<int, 0>{}, T + 2);
      issue_piece(std::integral_constant<int, 1>{}, T + 2);
#ifdef GEMM_SHALLOW
      asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
#else
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
#endif
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    LGKM0_BAR();
    MFMA_PAIR(1, w1f, 1, w0f, 0)
#undef MFMA_PAIR
  }
  TL_STAMP(2);
  if (group == 0) BAR();   // every wave must execute the same number of barriers
  TL_STAMP(3);
  // Every per-lane constant the next tile's set-up and the epilogue use (row / column offsets, pointers) derives from the lane
  // id.  Kept live ACROSS the K loop they cost registers the loop does not have (its 253 are accumulators + fragments + DMA
  // offsets) and the compiler spills them: ten scratch reloads, each behind its own `s_waitcnt vmcnt(0)`, were 3 us of every
  // tile (tools/gemm_timeline.py).  Re-reading the lane id here through an opaque instruction pair ends those live ranges at the
  // loop: everything below is recomputed from it (a few dozen VALU instructions per tile).
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
  st_r = lane >> 3; st_c = lane & 7;
  fr = lane & 15; fq = lane >> 4;

  // folded RMSNorm: this lane's eight row scales, requested before the next tile's set-up so that they land behind it
  float rs_v[8];
  if constexpr (!F8) {
    if (p.row_scale) {
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const int row = m0 + wr * 128 + m * 16 + fr;
        rs_v[m] = p.row_scale[gemm_map_row(row < p.M ? row : p.M - 1, p.a_group, p.a_gstride, p.a_off)];
      }
    }
  }
  // ---- next tile: its K-tile 0 goes into ring buffer 0 now; this tile's coordinates stay in em0/en0 for the epilogue ----
  const int em0 = m0, en0 = n0;
  const bool edirect = dir_tile;
  bid += gridDim.x;
  const bool has_next = bid < nwg;
  if (has_next) {
    set_tile(bid);
    issue_piece(std::integral_constant<int, 0>{}, 0);
    issue_piece(std::integral_constant<int, 1>{}, 0);
    issue_piece(std::integral_constant<int, 2>{}, 0);
    issue_piece(std::integral_constant<int, 3>{}, 0);
  }
  TL_STAMP(4);

  // ---- W8A8: dequantise the accumulators (per-row activation scale x per-output-channel weight scale, packed-row order) ----
  if constexpr (F8) {
    float swv[4][4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      int sc = wc * 64 + n * 16 + fq * 4;                                               // LDS row (n, fq*4 + e) -> its W row
      if (edirect) {
        if (EPI == VSTAR_EPI_SILU_MUL) { const int jo = fq * 8 + (n >> 1) * 4; sc = wc * 64 + (jo >> 4) * 32 + (n & 1) * 16 + (jo & 15); }
        else if (EPI == VSTAR_EPI_NONE && p.rope_cs != nullptr && en0 < p.rope_cols) sc = (wc >> 1) * 128 + (n >> 1) * 64 + (wc & 1) * 32 + fq * 8 + (n & 1) * 4;
        else sc = wc * 64 + (n >> 1) * 32 + fq * 8 + (n & 1) * 4;
      }
      const f32x4 t = *(const f32x4*)(p.w_scale + en0 + sc);                            // w_scale is padded like W
#pragma unroll
      for (int e = 0; e < 4; ++e) swv[n][e] = t[e];
    }
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int row = em0 + wr * 128 + m * 16 + fr;
      const float sa = p.a_scale[row < p.M ? row : p.M - 1];
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[m][n][e] *= sa * swv[n][e];
    }
  }

  // ---- RMSNorm folded into this linear: Linear(RMSNorm(x)) = rstd[row] * (x . (W * norm_w)^T) ----
  if constexpr (!F8) {
    if (p.row_scale) {
#pragma unroll
      for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[m][n][e] *= rs_v[m];
    }
  }

  // ---- epilogue ----
  if (!(p.debug_flags & 2)) {
  const int n_out = (EPI == VSTAR_EPI_SILU_MUL) ? p.N / 2 : p.N;
  if constexpr (OUT_F32) {
    // fp32 outputs (lm_head / head taps): direct accumulator-layout stores
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int row = em0 + wr * 128 + m * 16 + fr;
      if (row >= p.M) continue;
      const int64_t crow = gemm_map_row(row, p.c_group, p.c_gstride, p.c_off);
      if (EPI == VSTAR_EPI_SILU_MUL) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
          gemm_epilogue_store<EPI, OUT_F32>(p, crow, (en0 + wc * 64) / 2 + j * 16 + fq * 4, n_out, acc[m][2 * j], acc[m][2 * j + 1]);
      } else {
#pragma unroll
        for (int n = 0; n < 4; ++n)
          gemm_epilogue_store<EPI, OUT_F32>(p, crow, en0 + wc * 64 + n * 16 + fq * 4, n_out, acc[m][n], acc[m][n]);
      }
    }
  } else if (edirect) {
    // the in-register epilogue (gemm256_direct_epilogue.hpp, shared with gemm4w.hip): this wave IS one virtual wave (wr, wc)
    gemm256_direct_epilogue<EPI, !F8>(p, em0, en0, wr, wc, fr, fq, [&](auto mc, f32x4 (&a)[4]) {
      constexpr int m = decltype(mc)::value;
      a[0] = acc[m][0]; a[1] = acc[m][1]; a[2] = acc[m][2]; a[3] = acc[m][3];
    });
  } else {
    // bf16 outputs: bias in the accumulator layout, transpose through this wave's private LDS slab — 32 rows at a time, in
    // ring buffer 1 (dead: every wave is past the last barrier; buffer 0 is already receiving the next tile) — then
    // activation / residual and whole-line 16-B stores from a ROLLED loop (keeps the epilogue's code footprint small: it runs once per tile and an
    // unrolled 32-fragment epilogue with tail paths was ~10k instructions of cold i-cache).
    constexpr int WCOLS = (EPI == VSTAR_EPI_SILU_MUL) ? 32 : 64;     // output columns owned by this wave
    constexpr int NF = WCOLS / 16;                                    // 16-column fragments
    constexpr int RSTRIDE = WCOLS * 2 + 16;                           // padded LDS row (bytes)
    constexpr int CH = WCOLS / 8;                                     // 16-B chunks per row
    constexpr int RPI = 64 / CH;                                      // rows per wave-wide 16-B access
    char* slab = smem + TILE_BYTES + wave * (32 * (128 + 16));
    const int colbase = (EPI == VSTAR_EPI_SILU_MUL) ? (en0 + wc * 64) / 2 : en0 + wc * 64;
    // bias for this lane's 4-column groups (clamped: columns >= n_out are computed but never stored)
    float bias_v[NF][4];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      int bc = colbase + f * 16 + fq * 4;
      bc = bc + 4 <= n_out ? bc : (n_out - 4 > 0 ? n_out - 4 : 0);
      if (EPI != VSTAR_EPI_SILU_MUL && p.bias) {
        const lpx4 b = *(const lpx4*)(p.bias + bc);
#pragma unroll
        for (int e = 0; e < 4; ++e) bias_v[f][e] = lp2f((lp_t)b[e]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) bias_v[f][e] = 0.f;
      }
    }
    const int rl0 = lane / CH, ch = lane % CH;
    // fused RoPE (q|k column tiles of the LLaMA qkv projection): a head = 128 columns = the two neighbouring N-waves
    // (wc, wc^1); a lane's partner values (d +- 64) sit at the same slab position of the neighbour's slab, hence the
    // workgroup barriers around stage 2 (every wave executes them: the condition is tile-uniform).
    bool rope_tile = false;
    if constexpr (EPI == VSTAR_EPI_NONE) rope_tile = p.rope_cs != nullptr && en0 < p.rope_cols;
    const char* slab_partner = smem + TILE_BYTES + (wave ^ 1) * (32 * (128 + 16));
    // Residual (VSTAR_EPI_NONE, whole tiles in N): added in stage 1, in the accumulator layout — the quarter's eight 8-byte loads
    // are in flight together.  Added after the transpose (gemm_epilogue_store_row8) the load sits in the rolled row loop, one
    // memory round trip per 8 rows, 16 in series per tile (OWL-ViT out-proj 145 -> 130 us, fc2 336 -> 322, LLaMA o_proj 531 -> 518;
    // a further quarter of lookahead, or the whole tile's residual requested before the epilogue, measured the same / spilled).
    // Same arithmetic either way: rlp(rlp(acc + bias) + res).
    bool res1 = false;
    if constexpr (EPI == VSTAR_EPI_NONE && !F8)      // the fp8 kernels already spill: the extra registers cost them 10 % of the step
      res1 = p.res != nullptr && !rope_tile && en0 + BN <= n_out && (((uintptr_t)p.res & 7) == 0) && (p.ldr % 4 == 0);
    constexpr int NIT = 32 / RPI;
    const bool fast_tile = !F8 && !rope_tile && em0 + BM <= p.M && en0 + BN <= p.N && p.c_group <= 0 && (((uintptr_t)p.C & 15) == 0) &&
                           (p.ldc % 8 == 0) && !(p.debug_flags & 3) && (p.res == nullptr || res1);
    auto quarter_pass = [&](auto qc) {
      constexpr int qp = decltype(qc)::value;       // rows qp*32 .. +31 of the wave's 128-row tile
      lpx4 rv[2][NF];
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int row = em0 + wr * 128 + (qp * 2 + m) * 16 + fr;
        const lp_t* rp = p.res + gemm_map_row(row < p.M ? row : p.M - 1, p.c_group, p.c_gstride, p.c_off) * p.ldr + colbase + fq * 4;
#pragma unroll
        for (int f = 0; f < NF; ++f) rv[m][f] = res1 ? *(const lpx4*)(rp + f * 16) : (lpx4){0, 0, 0, 0};
      }
#pragma unroll
      for (int m = 0; m < 2; ++m) {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          lpx4 v;
          if (EPI == VSTAR_EPI_SILU_MUL) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              v[e] = (short)f2lp(act_silu_bf16(rlp(acc[qp * 2 + m][2 * f][e])) * rlp(acc[qp * 2 + m][2 * f + 1][e]));
          } else {   // stage 1 = bf16(acc + bias); the activation is applied after the transpose
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (short)f2lp(acc[qp * 2 + m][f][e] + bias_v[f][e]);
            if constexpr (EPI == VSTAR_EPI_NONE) {
              if (res1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (short)f2lp(lp2f((lp_t)v[e]) + lp2f((lp_t)rv[m][f][e]));
              }
            }
          }
          *(lpx4*)(slab + (m * 16 + fr) * RSTRIDE + (f * 16 + fq * 4) * 2) = v;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (fast_tile) {
        // interior tile, identity row map, aligned C, no RoPE: the rolled loop below spends ~100 instructions per row group on
        // 64-bit addresses, row maps and tail predicates (the epilogue is VALU-bound: 8 us per tile) — here one base pointer per
        // quarter pass, NIT unrolled {LDS read, activation, 16-byte store (+ the sum of squares of the stored values)}
        const int64_t row0 = em0 + wr * 128 + qp * 32 + rl0;
        lp_t* cq = (lp_t*)p.C + row0 * p.ldc + colbase + ch * 8;
        const int64_t cstep = (int64_t)RPI * p.ldc;
        lpx8 fv[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) fv[it] = *(const lpx8*)(slab + (it * RPI + rl0) * RSTRIDE + ch * 16);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          gemm_epilogue_act8<EPI>(fv[it]);
          if (p.sumsq_out) *(lpx8*)(cq + it * cstep) = fv[it];
          else __builtin_nontemporal_store(fv[it], (lpx8*)(cq + it * cstep));
        }
        if constexpr (EPI == VSTAR_EPI_NONE && !F8) {
          if (p.sumsq_out) {
            float* sq = p.sumsq_out + row0 * p.sumsq_ld + colbase / 64;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
              const float ss = gemm_sumsq_span64_chunks(fv[it]);
              if (ch == 0) sq[(int64_t)it * RPI * p.sumsq_ld] = ss;
            }
            if (p.stats_sum) {          // launch-uniform: the sums too (LayerNorm folded into the consumer)
#pragma unroll
              for (int it = 0; it < NIT; ++it) {
                const float sm = gemm_sum_span64_chunks(fv[it]);
                if (ch == 0) sq[(int64_t)it * RPI * p.sumsq_ld + p.stats_sum] = sm;
              }
            }
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        return;
      }
      if (EPI == VSTAR_EPI_NONE && rope_tile) BAR();
#pragma unroll 1
      for (int it = 0; it < 32 / RPI; ++it) {
        const int rl = it * RPI + rl0;
        lpx8 v = *(const lpx8*)(slab + rl * RSTRIDE + ch * 16);
        const int row = em0 + wr * 128 + qp * 32 + rl;
        if constexpr (EPI == VSTAR_EPI_NONE) {
          if (rope_tile) {
            const lpx8 vp = *(const lpx8*)(slab_partner + rl * RSTRIDE + ch * 16);
            int pos = (row < p.M ? row : p.M - 1) % p.rope_S;
            if (p.rope_R0 > 0 && pos >= p.rope_R0) pos = p.rope_Lc + ((pos - p.rope_R0) & 31);      // grouped sequences
            if (p.rope_tail > 0) pos = row >= p.rope_tail ? (row < p.M ? row : p.M - 1) - p.rope_tail : pos + p.rope_pos0;   // shared prefix
            const lpx8 c8 = *(const lpx8*)(p.rope_cs + (int64_t)pos * 128 + ch * 8);
            const lpx8 s8 = *(const lpx8*)(p.rope_cs + (int64_t)pos * 128 + 64 + ch * 8);
            const float sgn = (wc & 1) ? 1.0f : -1.0f;      // first half of the head: x*cos - partner*sin
#pragma unroll
            for (int e = 0; e < 8; ++e)
              v[e] = (short)f2lp(rlp(lp2f((lp_t)v[e]) * lp2f((lp_t)c8[e])) + rlp(sgn * lp2f((lp_t)vp[e]) * lp2f((lp_t)s8[e])));
          }
        }
        const int64_t crow = gemm_map_row(row < p.M ? row : p.M - 1, p.c_group, p.c_gstride, p.c_off);
        if (row < p.M && !(p.debug_flags & 1)) gemm_epilogue_store_row8<EPI>(p, crow, colbase + ch * 8, n_out, v, res1);
        if constexpr (EPI == VSTAR_EPI_NONE && !F8) {
          if (p.sumsq_out) {          // tile-uniform: statistics of the next RMSNorm from the values just stored (8 lanes = this row's 64 columns)
            const float ss = gemm_sumsq_span64_chunks(v);
            if (ch == 0 && row < p.M && colbase < n_out) p.sumsq_out[crow * p.sumsq_ld + colbase / 64] = ss;
            if (p.stats_sum) {
              const float sm = gemm_sum_span64_chunks(v);
              if (ch == 0 && row < p.M && colbase < n_out) p.sumsq_out[crow * p.sumsq_ld + p.stats_sum + colbase / 64] = sm;
            }
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (EPI == VSTAR_EPI_NONE && rope_tile) BAR();      // the neighbour has read this slab before it is rewritten
    };
    quarter_pass(std::integral_constant<int, 0>{});
    quarter_pass(std::integral_constant<int, 1>{});
    quarter_pass(std::integral_constant<int, 2>{});
    quarter_pass(std::integral_constant<int, 3>{});
  }
  }  // epilogue
  TL_STAMP(5);
  ++tl_tile;
  if (!has_next) break;
  }  // persistent tile loop
}

template <int EPI, bool OUT_F32, bool F8 = false>
hipError_t launch(const GemmParams& p, hipStream_t s) {
  if (gemm_plan_only()) return hipSuccess;
  static bool attr_done = false;
  auto kern = gemm256_kernel<EPI, OUT_F32, F8>;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  const int n_cu = gemm_device_cus();
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  hipLaunchKernelGGL(kern, dim3(tiles < n_cu ? tiles : n_cu), dim3(512), LDS_TOTAL, s, p);
  return hipGetLastError();
}

}  // namespace

#if defined(GEMM_TIMELINE) && !defined(VSTAR_LP_F16)
extern "C" int vstar_debug_gemm_timeline(unsigned long long* out) {     // 2 x 2 x 64 x 8 stamps
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_timeline), sizeof(g_timeline));
}
#endif

// Shapes this kernel accepts: K/64 even and >= 2, operands addressable with 32-bit element offsets.
bool gemm256_eligible(const GemmParams& p) {
  if (p.a_scale) {      // W8A8: K counts fp8 elements, a K-tile is 128 of them, K/128 even
    if (p.K % 256 != 0 || p.K < 256 || !p.w_scale) return false;
  }
  if (p.K % 128 != 0 || p.K < 128) return false;
  if (p.M < 1024 || p.N < 256) return false;
  const int64_t amax = (p.a_group > 0 ? ((int64_t)(p.M / p.a_group) + 1) * p.a_gstride + p.a_off + p.a_group : (int64_t)p.M) * p.lda;
  const int64_t npad = ((int64_t)p.N + BN - 1) / BN * BN;
  if (amax >= (1ll << 31) || npad * p.K >= (1ll << 31)) return false;
  return true;
}

hipError_t gemm256_lp(const GemmParams& p0, int epilogue, bool out_f32, hipStream_t s) {
  // VSTAR_GEMM_DIRECT=0 (A/B runs): the LDS-transposed epilogue everywhere (debug bit 2 = "no direct epilogue")
  static const bool no_direct = [] { const char* e = getenv("VSTAR_GEMM_DIRECT"); return e && atoi(e) == 0; }();
  GemmParams p = p0;
  if (no_direct) p.debug_flags |= 4;
  static const int stagger = [] { const char* e = getenv("VSTAR_GEMM_XCD_STAGGER"); return e ? atoi(e) & 0xff : 0; }();
  p.debug_flags |= stagger << 8;
  if (p.a_scale) {       // W8A8 instantiations: the two epilogues the LLaMA linears use
    if (out_f32) return hipErrorInvalidValue;
    if (epilogue == VSTAR_EPI_NONE) return launch<VSTAR_EPI_NONE, false, true>(p, s);
    if (epilogue == VSTAR_EPI_SILU_MUL) return launch<VSTAR_EPI_SILU_MUL, false, true>(p, s);
    return hipErrorInvalidValue;
  }
#define GEMM_CASE(E)                                                   \
  case E:                                                              \
    return out_f32 ? launch<E, true>(p, s) : launch<E, false>(p, s);
  switch (epilogue) {
    GEMM_CASE(VSTAR_EPI_NONE)
    GEMM_CASE(VSTAR_EPI_QUICK_GELU)
    GEMM_CASE(VSTAR_EPI_GELU)
    GEMM_CASE(VSTAR_EPI_RELU)
    GEMM_CASE(VSTAR_EPI_SILU_MUL)
  }
#undef GEMM_CASE
  return hipErrorInvalidValue;
}

}  // namespace VS_NS
