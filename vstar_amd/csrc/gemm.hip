// gemm.hip — bf16 MFMA GEMM for gfx950:  C[M,N] = epi(A[M,K] @ W[N,K]^T + bias) (+ residual)
//
// Replaces every nn.Linear / conv-as-GEMM on the VSM scoring path (SURVEY.md §8d shape list):
// CLIP/OWL-ViT qkv/out/fc1/fc2 (HF CLIPEncoderLayer via clip_encoder.py:53-57, owlvit.py:121-126),
// mm_projector (llava_arch.py:93-96), LLaMA q/k/v/o/gate/up/down (llava_llama.py:93-102),
// text_hidden_fcs_* (VSM.py:120-140), OWL-ViT class/box heads (owlvit.py:79-119), SAM head linears/convs.
//
// Design (CDNA4): 128x128x64 block tile, 4 waves (2x2), each wave a 64x64 sub-tile = 4x4 fragments of
// v_mfma_f32_16x16x32_bf16.  Both operands are K-contiguous ("NT"), staged HBM->LDS with 16-byte
// global_load_lds (no VGPR round trip), two LDS buffers, next tile's DMA issued before the current tile's
// MFMAs.  LDS rows are 128 B (8 x 16-B chunks); chunk index is XOR-swizzled with (row>>1)&7 so that every
// ds_read_b128 lane group touches 16 distinct 16-B slots (conflict-free).  global_load_lds writes
// lane-linear, so the swizzle is applied on the per-lane SOURCE address and again on the read address.
// MFMA operands are swapped (W fragment as "A", activation fragment as "B") so that each lane ends up
// with 4 consecutive output COLUMNS of one row -> 8-byte lpx4 / 16-byte f32x4 stores, and bias/residual
// loads of the same shape.  Workgroup ids are remapped XCD-aware (bijective) + grouped along M so that
// blocks sharing an L2 share weight columns.
#include "common.hpp"
#include "kernels.hpp"
#include "gemm_epilogue.hpp"
#include <cstdlib>

namespace VS_NS {

static thread_local bool t_plan_only = false;
static thread_local int t_plan_cus = 0;
static thread_local int t_last_mode = 0;
void gemm_set_plan(bool on, int cus) { t_plan_only = on; t_plan_cus = on ? cus : 0; }
bool gemm_plan_only() { return t_plan_only; }
int gemm_last_mode() { return t_last_mode; }

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int LDS_TILE = BM * BK * 2;            // 16 KiB per operand tile
constexpr int LDS_BUF = 2 * LDS_TILE;            // A + W = 32 KiB per K-tile
// STAGES = 2: double buffer (64 KiB, two workgroups per CU), every K-tile ends in __syncthreads (vmcnt(0) + barrier).
// STAGES = 3 (round 3, the default): 96-KiB ring, the DMA runs TWO K-tiles ahead behind a COUNTED vmcnt(8) and there is one
// bare s_barrier per K-tile — the latency regime of small batches (M = 640..2560: under-filled grids, one workgroup per CU,
// every weight byte straight from HBM) no longer pays a full DMA round trip per K-tile.  Same K order: bit-identical.

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ int64_t map_row(int r, int group, int64_t gstride, int64_t off) {
  return gemm_map_row(r, group, gstride, off);
}

#define BAR128()                                       \
  do {                                                 \
    __builtin_amdgcn_sched_barrier(0);                 \
    asm volatile("s_barrier" ::: "memory");            \
    __builtin_amdgcn_sched_barrier(0);                 \
  } while (0)

// NW = 64-column wave columns of the tile: 2 = the 128 x 128 tile (2 x 2 waves of 64 x 64), 1 = a 128 x 64 tile (4 x 1 waves of
// 32 x 64; loader-wave mode only, 72-KiB ring: TWO workgroups per CU) for under-filled grids — twice the workgroups, so every
// CU gets work and the two co-resident workgroups of a CU pull at 88 instead of 65 KB/us (tools/probes/dma_probe.hip).  A wave
// always owns whole 64-column spans, so the sumsq epilogue and the gate|up pairing are the same code.
// NW = 4: a 128 x 256 tile (2 x 4 = EIGHT compute waves of 64 x 64 + the loader wave, 144-KiB ring, one workgroup per CU) for
// the wide projections of small batches (qkv / gate|up at M = 640: 480 / 860 tiles of 128^2 on 256 CUs): three quarters of the
// operand bytes per MFMA of the 128^2 tile, which is what bounds this regime (see launch<> below).
template <int EPI, bool OUT_F32, int MODE, int NW = 2>
__global__ __launch_bounds__(MODE == 5 ? (NW == 4 ? 576 : 320) : 256, (NW != 4 && (MODE == 2 || NW == 1)) ? 2 : 1) void gemm128_kernel(const GemmParams p) {
  constexpr bool LDR = MODE == 5;
  constexpr int STAGES = LDR ? 3 : MODE;
  static_assert(NW == 2 || LDR, "the 128 x 64 and 128 x 256 tiles exist in loader-wave mode only");
  constexpr int BN = 64 * NW;                      // shadows the file-level BN (= 128)
  constexpr int MF = NW == 1 ? 2 : 4;              // 16-row fragments per wave
  constexpr int NCW = NW == 4 ? 8 : 4;             // compute waves; the loader wave is wave NCW
  constexpr int W_TILE = BN * BK * 2;
  constexpr int LDS_BUF = LDS_TILE + W_TILE;       // shadows the file-level LDS_BUF (= 32 KiB)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- tile id: XCD-aware bijective remap, then GROUP_M ordering ----
  const int tiles_m = (p.M + BM - 1) / BM;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int nwg = tiles_m * tiles_n;
  int t;
  {
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  constexpr int GROUP_M = 8;
  const int in_group = GROUP_M * tiles_n;
  const int grp = t / in_group;
  const int first_m = grp * GROUP_M;
  const int gsz = min(tiles_m - first_m, GROUP_M);
  const int rem = t - grp * in_group;
  const int tm = first_m + rem % gsz;
  const int tn = rem / gsz;
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- per-lane staging sources (4 x 1-KiB DMA pieces per operand per wave) ----
  const int st_r = lane >> 3;      // row within the 8-row piece
  const int st_c = lane & 7;       // 16-B chunk within the 128-B LDS row
  const lp_t* a_src[4];
  const lp_t* w_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (i * 4 + wave) * 8 + st_r;          // LDS row 0..127
    const int cg = st_c ^ ((r >> 1) & 7);             // global chunk that lives at LDS chunk st_c
    int ar = m0 + r;
    ar = ar < p.M ? ar : p.M - 1;                     // clamp: rows >= M are computed but never stored
    a_src[i] = p.A + map_row(ar, p.a_group, p.a_gstride, p.a_off) * p.lda + cg * 8;
    w_src[i] = p.W + (int64_t)(n0 + r) * p.K + cg * 8;   // W is padded to a multiple of BN rows
  }

  auto stage = [&](int buf, int k0) {
    char* base = smem + buf * LDS_BUF;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int piece = (i * 4 + wave) * 1024;
      __builtin_amdgcn_global_load_lds((gptr_t)(a_src[i] + k0), (lptr_t)(base + piece), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(w_src[i] + k0), (lptr_t)(base + LDS_TILE + piece), 16, 0, 0);
    }
  };

  if constexpr (LDR) {
    if (wave == NCW) {
      // ---- loader wave: all 16 A pieces + BN / 8 W pieces of every K-tile, two K-tiles ahead of the compute waves ----
      constexpr int WP = BN / 8;                     // 1-KiB pieces of the W tile
      constexpr int NP = WP > 16 ? WP : 16;
      const lp_t* a_all[16];
      const lp_t* w_all[WP];
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        const int r = j * 8 + st_r;
        const int cg = st_c ^ ((r >> 1) & 7);
        if (j < 16) {
          int ar = m0 + r;
          ar = ar < p.M ? ar : p.M - 1;
          a_all[j] = p.A + map_row(ar, p.a_group, p.a_gstride, p.a_off) * p.lda + cg * 8;
        }
        if (j < WP) w_all[j] = p.W + (int64_t)(n0 + r) * p.K + cg * 8;
      }
      auto issue = [&](int buf, int k0) {
        char* base = smem + buf * LDS_BUF;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
          if (j < 16) __builtin_amdgcn_global_load_lds((gptr_t)(a_all[j] + k0), (lptr_t)(base + j * 1024), 16, 0, 0);
          if (j < WP) __builtin_amdgcn_global_load_lds((gptr_t)(w_all[j] + k0), (lptr_t)(base + LDS_TILE + j * 1024), 16, 0, 0);
        }
      };
      const int nk = p.K / BK;
      issue(0, 0);
      if (nk > 1) issue(1, BK);
      int slot = 2;
      for (int kt = 0; kt < nk; ++kt) {
        __builtin_amdgcn_sched_barrier(0);
        // K-tile kt has landed, the 16 + BN / 8 pieces of kt+1 may be in flight
        if (kt + 1 >= nk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if constexpr (NW == 4) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
        else if constexpr (NW == 2) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        BAR128();                 // publishes K-tile kt; every compute wave is done with slot (kt-1) % 3
        if (kt + 2 < nk) issue(slot, (kt + 2) * BK);
        slot = slot == 2 ? 0 : slot + 1;
      }
      return;
    }
  }

  // ---- fragment read offsets ----
  const int wr = NW == 4 ? wave >> 2 : (NW == 2 ? wave >> 1 : wave), wc = NW == 4 ? wave & 3 : (NW == 2 ? wave & 1 : 0);
  const int fr = lane & 15, fq = lane >> 4;
  const int swz = (fr >> 1) & 7;
  int a_rd[2], w_rd[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int ch = ((kk * 4 + fq) ^ swz) * 16;
    a_rd[kk] = (wr * (MF * 16) + fr) * 128 + ch;
    w_rd[kk] = LDS_TILE + (wc * 64 + fr) * 128 + ch;
  }

  f32x4 acc[MF][4];
#pragma unroll
  for (int m = 0; m < MF; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nkt = p.K / BK;
  if constexpr (!LDR) stage(0, 0);
  if constexpr (LDR) {
  } else if constexpr (STAGES == 2) {
    __syncthreads();   // drains the DMA (vmcnt(0)) and makes tile 0 visible
  } else {
#pragma unroll
    for (int t = 1; t < STAGES - 1; ++t)
      if (t < nkt) stage(t, t * BK);
  }

  int cur = 0;         // ring slot of K-tile kt
  for (int kt = 0; kt < nkt; ++kt) {
    if constexpr (LDR) {
      BAR128();          // the loader wave has K-tile kt in slot `cur`
    } else if constexpr (STAGES == 2) {
      cur = kt & 1;
      if (kt + 1 < nkt) stage(cur ^ 1, (kt + 1) * BK);
    } else {
      // this wave's 8 DMA pieces of K-tile kt have landed once at most the 8 of K-tile kt+1 are still in flight (vmcnt retires
      // in order); the barrier then makes every wave's pieces visible AND proves that every wave is done reading slot
      // (kt-1) % STAGES (its fragment reads were consumed by MFMAs issued before this point) — the slot K-tile kt+STAGES-1 goes
      // into.
      __builtin_amdgcn_sched_barrier(0);
      if (kt + 1 < nkt) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      BAR128();
      if (kt + STAGES - 1 < nkt) stage(cur >= 1 ? cur - 1 : STAGES - 1, (kt + STAGES - 1) * BK);      // (cur + STAGES - 1) % STAGES
      __builtin_amdgcn_sched_barrier(0);
    }
    const char* base = smem + cur * LDS_BUF;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      lpx8 af[MF], wf[4];
#pragma unroll
      for (int m = 0; m < MF; ++m) af[m] = *(const lpx8*)(base + a_rd[kk] + m * 2048);
#pragma unroll
      for (int n = 0; n < 4; ++n) wf[n] = *(const lpx8*)(base + w_rd[kk] + n * 2048);
#pragma unroll
      for (int m = 0; m < MF; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
          acc[m][n] = mfma_16x16x32(wf[n], af[m], acc[m][n]);
    }
    if constexpr (!LDR && STAGES == 2) {
      __syncthreads();   // next tile landed (vmcnt(0)) + everyone done reading buf[cur]
    } else {
      cur = cur == STAGES - 1 ? 0 : cur + 1;
    }
  }

  // ---- epilogue: lane owns row (m*16+fr), columns fq*4..fq*4+3 of each 16x16 fragment ----
  const int n_out = (EPI == VSTAR_EPI_SILU_MUL) ? p.N / 2 : p.N;
  if (p.row_scale) {       // RMSNorm folded into this linear (see kernels.hpp)
#pragma unroll
    for (int m = 0; m < MF; ++m) {
      const int row = m0 + wr * (MF * 16) + m * 16 + fr;
      const float rs = p.row_scale[map_row(row < p.M ? row : p.M - 1, p.a_group, p.a_gstride, p.a_off)];
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[m][n][e] *= rs;
    }
  }
  if constexpr (EPI == VSTAR_EPI_NONE && !OUT_F32) {
    if (p.sumsq_out) {     // block-uniform: also write the 64-column sums of squares of the stored values (no early exits: shuffles)
#pragma unroll
      for (int m = 0; m < MF; ++m) {
        const int row = m0 + wr * (MF * 16) + m * 16 + fr;
        const int64_t crow = map_row(row < p.M ? row : p.M - 1, p.c_group, p.c_gstride, p.c_off);
        float o[4][4];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          if (row < p.M) gemm_epilogue_store<EPI, OUT_F32>(p, crow, n0 + wc * 64 + n * 16 + fq * 4, n_out, acc[m][n], acc[m][n], o[n]);
          else o[n][0] = o[n][1] = o[n][2] = o[n][3] = 0.f;
        }
        const float ss = gemm_sumsq_span64_frags(o);
        if (fq == 0 && row < p.M && n0 + wc * 64 < n_out) p.sumsq_out[crow * p.sumsq_ld + (n0 + wc * 64) / 64] = ss;
        if (p.stats_sum) {       // launch-uniform
          const float sm = gemm_sum_span64_frags(o);
          if (fq == 0 && row < p.M && n0 + wc * 64 < n_out) p.sumsq_out[crow * p.sumsq_ld + p.stats_sum + (n0 + wc * 64) / 64] = sm;
        }
      }
      return;
    }
  }
#pragma unroll
  for (int m = 0; m < MF; ++m) {
    const int row = m0 + wr * (MF * 16) + m * 16 + fr;
    if (row >= p.M) continue;
    const int64_t crow = map_row(row, p.c_group, p.c_gstride, p.c_off);
    if (EPI == VSTAR_EPI_SILU_MUL) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
        gemm_epilogue_store<EPI, OUT_F32>(p, crow, (n0 + wc * 64) / 2 + j * 16 + fq * 4, n_out, acc[m][2 * j], acc[m][2 * j + 1]);
    } else {
#pragma unroll
      for (int n = 0; n < 4; ++n)
        gemm_epilogue_store<EPI, OUT_F32>(p, crow, n0 + wc * 64 + n * 16 + fq * 4, n_out, acc[m][n], acc[m][n]);
    }
  }
}

}  // namespace
int gemm_device_cus();
namespace {

template <int EPI, bool OUT_F32, int MODE, int NW = 2>
hipError_t launch_stages(const GemmParams& p, hipStream_t s) {
  t_last_mode = MODE == 5 ? (NW == 4 ? 7 : (NW == 1 ? 6 : 5)) : MODE;
  if (t_plan_only) return hipSuccess;
  static bool attr_done = false;
  auto kern = gemm128_kernel<EPI, OUT_F32, MODE, NW>;
  constexpr int bn = 64 * NW;
  constexpr int lds = (MODE == 5 ? 3 : MODE) * (LDS_TILE + bn * BK * 2);
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + bn - 1) / bn);
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(MODE == 5 ? (NW == 4 ? 576 : 320) : 256), lds, s, p);
  return hipGetLastError();
}

template <int EPI, bool OUT_F32>
hipError_t launch(const GemmParams& p, hipStream_t s) {
  // Under-filled grids (at most one tile per CU: the small-batch o_proj / down_proj, CLIP out / fc2) take the loader-wave ring
  // (MODE 5): one workgroup per CU anyway, so the 96-KiB ring costs no occupancy.  Fuller grids keep the 64-KiB double buffer,
  // where two co-resident workgroups per CU hide each other's DMA round trips and a ragged last round costs nothing.
  // Measured (profiles/r03_gemm_small_batch.txt, M = 640): o_proj 65 -> 40 us, down_proj 169 -> 104 us, CLIP fc2 54 -> 37 us;
  // with more tiles than CUs the double buffer wins by 5-15 %.  What bounds both: a CU pulls operand tiles from L2 at
  // ~65 (one workgroup) to ~88 KB/us (two) whatever the instruction (tools/probes/dma_probe.hip), and a 128^2 tile needs 32 KiB
  // per K-tile.  VSTAR_GEMM128_STAGES=2|5|6 forces one variant (A/B runs).  Same K order in both: bit-identical results.
  static const int force = [] { const char* e = getenv("VSTAR_GEMM128_STAGES"); return e ? atoi(e) : 0; }();
  const int64_t tiles = (int64_t)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  // 6 = loader-wave ring on the 128 x 64 tile: grids that fill less than HALF the CUs (CLIP / OWL-ViT out-proj and fc2 at small
  // batches: 40 - 114 tiles) — twice the workgroups beats the extra A traffic (12 vs 18 us, 27 vs 37 us at M = 577); between half
  // and all CUs (LLaMA o_proj / down_proj at M = 640: 160 tiles, K up to 11008) the 128 x 128 tile moves fewer bytes and wins
  // (40 vs 44 us, 104 vs 125 us).  5 = loader-wave ring on the 128 x 128 tile.
  const int64_t cus = gemm_device_cus();
  // 7 = loader-wave ring on the 128 x 256 tile (W is padded to 256 rows by contract).
  // More 128^2 tiles than CUs and a long K (the LLaMA linears of 1 - 3 crops: qkv / gate|up at M = 640, o / down at M = 1280 and
  // 1920): whole rounds of 128 x 256 tiles at 1.05 us per K-tile against whole rounds of two co-resident 128^2 tiles at 1.28 us
  // (profiles/r03_gemm_small_batch.txt: qkv 79 -> 68 us, gate|up 144 -> 126 us at M = 640; o 75 -> 65 us, down 183 -> 171 us at
  // M = 1280).  Short-K shapes (the ViT towers) lose on the wide tile's prologue and direct-store epilogue and keep mode 2.
  const int64_t tiles7 = (int64_t)((p.M + BM - 1) / BM) * ((p.N + 255) / 256);
  const bool wide = p.K >= 2048 && tiles > cus &&
                    1.05 * (double)((tiles7 + cus - 1) / cus) <= 1.28 * (double)((tiles + 2 * cus - 1) / (2 * cus));
  const int mode = force == 2 || force == 5 || force == 6 || force == 7 ? force
                   : (p.K < 4 * BK ? 2 : (wide ? 7 : (tiles > cus ? 2 : (2 * tiles <= cus ? 6 : 5))));
  if (mode == 7) return launch_stages<EPI, OUT_F32, 5, 4>(p, s);
  if (mode == 6) return launch_stages<EPI, OUT_F32, 5, 1>(p, s);
  if (mode == 5) return launch_stages<EPI, OUT_F32, 5, 2>(p, s);
  return launch_stages<EPI, OUT_F32, 2, 2>(p, s);
}

}  // namespace

bool gemm256_eligible(const GemmParams& p);
hipError_t gemm256_lp(const GemmParams& p, int epilogue, bool out_f32, hipStream_t s);
bool gemm4w_eligible(const GemmParams& p, int epilogue, bool out_f32);      // gemm4w.hip: the 4-wave / AGPR 256^2 kernel
hipError_t gemm4w_lp(const GemmParams& p, int epilogue, hipStream_t s);

static thread_local int t_last_tile = 0;
int gemm_last_tile() { return t_last_tile; }

int gemm_device_cus() {
  if (t_plan_only && t_plan_cus > 0) return t_plan_cus;
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 8;
    n_cu = prop.multiProcessorCount / 8 * 8;
    if (n_cu <= 0) n_cu = 8;
  }
  return n_cu;
}

hipError_t gemm_lp(const GemmParams& p, int epilogue, bool out_f32, hipStream_t s) {
  t_last_tile = 0;
  t_last_mode = 0;
  if (p.M <= 0 || p.N <= 0) return hipSuccess;
  if (p.K % BK != 0 || p.K <= 0 || p.norm_w) return hipErrorInvalidValue;   // fused RMSNorm: skinny kernel only
  if (((uintptr_t)p.A & 15) || ((uintptr_t)p.W & 15) || (p.lda % 8)) return hipErrorInvalidValue;
  static const int env_force = [] { const char* e = getenv("VSTAR_GEMM_TILE"); return e ? atoi(e) : 0; }();
  const int force = p.tile_force ? p.tile_force : env_force;
  if (force != 0 && force != 128 && force != 256 && force != GEMM_TILE_4W) return hipErrorInvalidValue;
  if (p.a_mx || p.c_mx) {      // block-scaled W8A8 (mx.hpp) exists in the 4-wave kernel only: callers check gemm_mx_supported
    if (out_f32 || (force != 0 && force != GEMM_TILE_4W) || !gemm4w_eligible(p, epilogue, out_f32)) return hipErrorInvalidValue;
    t_last_tile = GEMM_TILE_4W;
    return gemm4w_lp(p, epilogue, s);
  }
  const bool elig = gemm256_eligible(p);
  if (p.tile_force == 256 && !elig) return hipErrorInvalidValue;   // an explicit per-call request must not be silently re-routed
  // The 4-wave / AGPR kernel (gemm4w.hip) takes the launches made of interior 256^2 tiles with the in-register epilogue — WHERE the
  // dispatcher would run the 256^2 kernel at all (below: under-filled grids still go to the 128^2 family, ragged rounds are still
  // split).  Same k order and epilogue arithmetic as gemm256: bit-identical (tests/test_ops_gpu.py::test_gemm4w_equals_gemm256).
  // tile_force = 256 keeps the 8-wave kernel (A/B, tests), GEMM_TILE_4W demands this one; VSTAR_GEMM4W=0/1 sets the default.
  static const int env4w = [] { const char* e = getenv("VSTAR_GEMM4W"); return e ? atoi(e) : GEMM4W_DEFAULT; }();
  if (force == GEMM_TILE_4W) {
    if (!(elig && gemm4w_eligible(p, epilogue, out_f32))) return hipErrorInvalidValue;
    t_last_tile = GEMM_TILE_4W;
    return gemm4w_lp(p, epilogue, s);
  }
  auto big = [&](const GemmParams& q) -> hipError_t {       // the 256^2 launch of this call (whole, or the leading part of a split)
    if (force == 0 && env4w != 0 && gemm4w_eligible(q, epilogue, out_f32)) {
      t_last_tile = GEMM_TILE_4W;
      return gemm4w_lp(q, epilogue, s);
    }
    t_last_tile = 256;
    return gemm256_lp(q, epilogue, out_f32, s);
  };
  if (force != 128 && elig) {   // W is padded to 256 rows
    // Under-filled grids (small batches: e.g. o_proj at 1280 rows = 80 tiles of 256^2 on 256 CUs): the 128^2 kernel has four
    // times the tiles; one of its tiles takes ~0.36 of a 256^2 tile (1/4 of the work at ~0.7 of the efficiency), so compare
    // whole rounds over the CUs.  Both kernels accumulate K in the same order: results are bit-identical either way
    // (tests/test_ops_gpu.py::test_gemm128_equals_gemm256).
    const int64_t cus = gemm_device_cus();
    const int64_t t256 = (int64_t)((p.M + 255) / 256) * ((p.N + 255) / 256), t128 = (int64_t)((p.M + 127) / 128) * ((p.N + 127) / 128);
    const bool prefer128 = force != 256 && !p.rope_cs && !p.a_scale && t256 < cus &&
                           0.36 * (double)((t128 + cus - 1) / cus) < (double)((t256 + cus - 1) / cus);
    if (!prefer128) {
      // Ragged last round (e.g. CLIP out-proj / fc2: 18464 x 1024 = 292 tiles = one full round + 36 tiles on 256 CUs): when the
      // 256^2 grid ends with a thinly filled round, the rows of that round go to the 128^2 kernel instead — whole rounds of
      // 256^2 tiles over the leading rows, then the trailing rows as ONE round of 128^2 tiles (CLIP fc2 172 -> 155 us, out-proj
      // 65 -> 60 us; with two or more 128^2 rounds the split measured no better than the ragged round).  Bit-identical
      // to the unsplit launch (same K order in both kernels).  Only for identity row maps (the tail is addressed by pointer
      // offset) and without fused RoPE / W8A8 (row-indexed side inputs).
      const int64_t rt = (p.M + 255) / 256, ct = (p.N + 255) / 256;
      const int64_t full = t256 / cus, rem = t256 - full * cus;
      static const bool env_split = [] { const char* e = getenv("VSTAR_GEMM_SPLIT"); return !e || atoi(e) != 0; }();   // A/B runs
      const bool split_ok = env_split && force == 0 && p.a_group <= 0 && p.c_group <= 0 && !p.rope_cs && !p.a_scale && !p.debug_flags;
      if (split_ok && full >= 1 && rem > 0) {
        const int64_t rt1 = full * cus / ct;                       // leading row tiles: at most `full` rounds of 256^2 tiles
        const int64_t M1 = rt1 * 256, M2 = p.M - M1;
        const int64_t t128b = ((M2 + 127) / 128) * ((p.N + 127) / 128);
        if (rt1 >= 4 && rt1 < rt && M2 > 0 && t128b <= cus) {      // ONE round of 128^2 tiles (measured: two rounds no longer pay)
          GemmParams a = p, b = p;
          a.M = (int)M1;
          b.M = (int)M2;
          b.A = p.A + M1 * p.lda;
          b.C = out_f32 ? (void*)((float*)p.C + M1 * p.ldc) : (void*)((lp_t*)p.C + M1 * p.ldc);
          if (p.res) b.res = p.res + M1 * p.ldr;
          if (p.row_scale) b.row_scale = p.row_scale + M1;
          if (p.sumsq_out) b.sumsq_out = p.sumsq_out + M1 * p.sumsq_ld;      // (stats_sum is an offset inside a partial row: unchanged)
          b.tile_force = 128;
          hipError_t e = big(a);
          if (e != hipSuccess) return e;
          e = gemm_lp(b, epilogue, out_f32, s);
          t_last_tile = 256 + 128;                                 // observable: split launch
          return e;
        }
      }
      return big(p);
    }
  }
  if (p.rope_cs || p.a_scale) return hipErrorInvalidValue;   // fused RoPE / W8A8 exist only in the 256^2 kernel: callers check gemm256_eligible
  t_last_tile = 128;
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
  t_last_tile = 0;
  return hipErrorInvalidValue;
}

}  // namespace VS_NS
