// decode.hip — kernels of the KV-cached language-model path (VQA-LLM: LLaVA/llava/model/language_model/
// llava_search_llama.py:56-113 driven by vstar_bench_eval.py:78-165; HF LlamaAttention 4.31 with past_key_values) and of the
// object-feature Perceiver resampler (LLaVA/llava/model/multimodal_projector/perceiver.py:25-121).
//
//   gemm_skinny_kernel   out[M<=64, N] = A · W^T for decode-sized M: HBM-bound weight streaming.  One workgroup owns 16 (or
//                        16 gate + 16 up) output columns, its 8 waves split K in an interleaved fashion so that the
//                        workgroup as a whole reads 512 contiguous bytes of every W row per step; partial sums meet in LDS.
//   rope_kv_append       rotate-half RoPE (HF rounding points) on q,k in place at per-row absolute positions + K/V rows
//                        written into the per-slot cache [slot][head][ctx][128].
//   cached_attn_kernel   one workgroup per (new row, head): scores against the cached keys (prefix slot below `past`, own
//                        slot from there on — option scoring forks a shared question prefix without copying it), fp32
//                        softmax, probabilities rounded to the storage type like HF, PV from the cached values.
//   perceiver_attn       32 latents x (256 media + 32 latent) keys, 16 heads x 96 dims: one wave per (image, head, latent).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdlib>
#include <type_traits>
#include "common.hpp"
#include "kernels.hpp"
#include "gemm_epilogue.hpp"

namespace VS_NS {

namespace {

// ------------------------------------------------ skinny GEMM ------------------------------------------------
// Operands as in the big kernels: W fragment is the MFMA A operand (16 output columns x 32 k), the activation fragment
// the B operand (16 rows x 32 k); lane (fr = lane%16, g = lane/16) loads 16 bytes at k = ks*32 + g*8 of W row / A row fr.
// The accumulator lane then owns output row fr, columns 4g..4g+3 — the layout gemm_epilogue_store expects.
constexpr int SK_WAVES = 8;
constexpr int SKR_WAVE_BYTES = 10240;      // gemm_skinny_ring_kernel: LDS ring per wave

template <int EPI, bool OUT_F32, int MT>
__global__ __launch_bounds__(SK_WAVES * 64) void gemm_skinny_kernel(const GemmParams p) {
  constexpr int NT = (EPI == VSTAR_EPI_SILU_MUL) ? 2 : 1;
  __shared__ float red[SK_WAVES][MT][64][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 16 * NT;

  const lp_t* wp[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) wp[t] = p.W + (int64_t)(n0 + t * 16 + fr) * p.K + g * 8;
  const lp_t* ap[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    int row = m * 16 + fr;
    row = row < p.M ? row : p.M - 1;          // rows past M repeat the last row (their results are never stored)
    ap[m] = p.A + (int64_t)row * p.lda + g * 8;
  }
  f32x4 acc[NT][MT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[t][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- optional fused RMSNorm of the A rows: row statistics first (same summation order as norm_kernel<true>) ----
  const bool fuse_norm = p.norm_w != nullptr;
  float rstd[MT];
  const lp_t* nwp = p.norm_w + g * 8;
  if (fuse_norm) {
    float* rs_sh = &red[0][0][0][0];
    for (int row = wave; row < p.M; row += SK_WAVES) {
      const lp_t* xr = p.A + (int64_t)row * p.lda;
      float sum = 0.f;
      for (int vi = lane; vi * 8 < p.K; vi += 64) {
        const lpx8 t = *(const lpx8*)(xr + vi * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float v = lp2f((lp_t)t[e]);
          sum += v * v;
        }
      }
      sum = wave_sum(sum);
      if (lane == 0) rs_sh[row] = rsqrtf(sum / (float)p.K + p.norm_eps);
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int row = m * 16 + fr;
      rstd[m] = rs_sh[row < p.M ? row : p.M - 1];
    }
    __syncthreads();           // rs_sh aliases the reduction buffer used below
  }
  auto normed = [&](lpx8 x, lpx8 w, float rs) {
    lpx8 y;
#pragma unroll
    for (int e = 0; e < 8; ++e) y[e] = (short)f2lp(lp2f((lp_t)w[e]) * rlp(lp2f((lp_t)x[e]) * rs));
    return y;
  };

  // K is walked in DOUBLE steps of 64 elements (two MFMA k-steps = one whole 128-byte line of every W row), interleaved
  // over the 8 waves; UH double steps (2*UH k-steps) are in flight per wave before the first MFMA consumes them.
  constexpr int UH = (MT == 2) ? 2 : (MT == 1 ? 1 : 2);
  const int nd = p.K >> 6;
  int ds = wave;
  for (; ds + (UH - 1) * SK_WAVES < nd; ds += UH * SK_WAVES) {
    lpx8 wf[2 * UH][NT], af[2 * UH][MT];
#pragma unroll
    for (int u = 0; u < 2 * UH; ++u) {
      const int k = (ds + (u >> 1) * SK_WAVES) * 64 + (u & 1) * 32;
#pragma unroll
      for (int t = 0; t < NT; ++t) wf[u][t] = __builtin_nontemporal_load((const lpx8*)(wp[t] + k));   // streamed once
#pragma unroll
      for (int m = 0; m < MT; ++m) af[u][m] = *(const lpx8*)(ap[m] + k);
    }
    if (fuse_norm) {
#pragma unroll
      for (int u = 0; u < 2 * UH; ++u) {
        const lpx8 nw = *(const lpx8*)(nwp + (ds + (u >> 1) * SK_WAVES) * 64 + (u & 1) * 32);
#pragma unroll
        for (int m = 0; m < MT; ++m) af[u][m] = normed(af[u][m], nw, rstd[m]);
      }
    }
#pragma unroll
    for (int u = 0; u < 2 * UH; ++u)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[t][m] = mfma_16x16x32(wf[u][t], af[u][m], acc[t][m]);
  }
  for (; ds < nd; ds += SK_WAVES) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k = ds * 64 + h * 32;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const lpx8 wf = *(const lpx8*)(wp[t] + k);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          lpx8 a = *(const lpx8*)(ap[m] + k);
          if (fuse_norm) a = normed(a, *(const lpx8*)(nwp + k), rstd[m]);
          acc[t][m] = mfma_16x16x32(wf, a, acc[t][m]);
        }
      }
    }
  }
  // ---- cross-wave reduction (fixed order => results do not depend on scheduling); wave m finishes row tile m ----
  f32x4 s[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (t) __syncthreads();
#pragma unroll
    for (int m = 0; m < MT; ++m) *(f32x4*)red[wave][m][lane] = acc[t][m];
    __syncthreads();
    if (wave < MT) {
      s[t] = *(const f32x4*)red[0][wave][lane];
#pragma unroll
      for (int w = 1; w < SK_WAVES; ++w) s[t] += *(const f32x4*)red[w][wave][lane];
    }
  }
  if (wave >= MT) return;
  const int n_out = (EPI == VSTAR_EPI_SILU_MUL) ? p.N / 2 : p.N;
  const int row = wave * 16 + fr;
  if (row >= p.M) return;
  if (EPI == VSTAR_EPI_SILU_MUL) gemm_epilogue_store<EPI, OUT_F32>(p, row, n0 / 2 + g * 4, n_out, s[0], s[NT - 1]);
  else gemm_epilogue_store<EPI, OUT_F32>(p, row, n0 + g * 4, n_out, s[0], s[0]);
}

// ------------------------------------------------ skinny GEMM, M <= 8, operands through LDS rings ------------------------
// The same arithmetic as gemm_skinny_kernel<EPI, OUT_F32, 1> — the same MFMA operands in the same order per wave (wave w owns
// the 64-element double steps ds = w, w + 8, ...), the same fixed-order cross-wave reduction: BIT-IDENTICAL results — but no
// operand passes through registers on its way in.  Every wave owns a private ring of RD stages in LDS; a stage is one double
// step: the workgroup's 16 (x NT) weight rows (2 KiB each, two LDS-DMA requests of 8 rows x 128 B: whole cache lines, where
// the register loads above touch 16 rows x 64 B per request) plus ONE 1-KiB request for the activation side — rows 0..6 = the
// (up to 7) activation rows, row 7 = the RMSNorm weight slice when the norm is fused.  Requests cost no registers, so
// RD stages (9 - 10 KiB per wave, 72 - 80 KiB per workgroup, two workgroups per CU) are in flight instead of 16 KiB per
// workgroup: a decode step at batch 1 has one workgroup per CU in o_proj / down_proj, and 16 KiB in flight per CU is a third of
// what 6 TB/s x ~2 us of latency needs.  No barrier in the K loop (private rings, counted vmcnt).  Everything in the queue is an
// LDS-DMA request on purpose: register loads mixed into the counted waits returned out of order with the DMA requests
// (wrong results under load), requests of one kind retire in order.
template <int EPI, bool OUT_F32, bool NORM>
__global__ __launch_bounds__(SK_WAVES * 64, 2) void gemm_skinny_ring_kernel(const GemmParams p) {
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  constexpr int NT = (EPI == VSTAR_EPI_SILU_MUL) ? 2 : 1;
  constexpr int RD = NT == 1 ? 3 : 2;                 // ring depth (stages)
  constexpr int STAGE = NT * 2048 + 1024;             // bytes per stage: W tiles | activation piece
  constexpr int SOPS = 2 * NT + 1;                    // DMA requests per stage and wave
  extern __shared__ __attribute__((aligned(16))) char smem[];      // [8 waves][10 KiB]; reused for the reduction
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 16 * NT;
  char* ring = smem + wave * SKR_WAVE_BYTES;

  // DMA sources of this lane: a request moves 8 rows x 128 B, lane -> row lane/8, LDS slot lane%8 <- global chunk slot ^ ((row>>1)&7)
  const int st_r = lane >> 3, st_c = lane & 7;
  const lp_t* wsrc[NT][2];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int row = h * 8 + st_r;
      wsrc[t][h] = p.W + (int64_t)(n0 + t * 16 + row) * p.K + (st_c ^ ((row >> 1) & 7)) * 8;
    }
  const lp_t* asrc;
  {
    const int cg = (st_c ^ ((st_r >> 1) & 7)) * 8;
    const int ar = st_r < p.M ? st_r : p.M - 1;
    asrc = (NORM && st_r == 7) ? p.norm_w + cg : p.A + (int64_t)ar * p.lda + cg;
  }
  // fragment reads: W row fr, chunk (u*4 + g) ^ ((fr>>1)&7); activation row min(fr, M-1) (< 8); norm weights = row 7
  const int arow = fr < p.M ? fr : p.M - 1;
  int w_rd[2], a_rd[2], n_rd[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    w_rd[u] = fr * 128 + (((u * 4 + g) ^ ((fr >> 1) & 7)) * 16);
    a_rd[u] = NT * 2048 + arow * 128 + (((u * 4 + g) ^ ((arow >> 1) & 7)) * 16);
    n_rd[u] = NT * 2048 + 7 * 128 + (((u * 4 + g) ^ 3) * 16);
  }

  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nd = p.K >> 6;
  const int n = (nd - wave + SK_WAVES - 1) / SK_WAVES;          // this wave's double steps: ds = wave + 8 i
  // tile-major weights (GemmParams::W_tiled): this workgroup's pieces lie back to back, [k-step][t][h] x 1 KiB, already in request order
  const bool tiled = p.W_tiled != nullptr;
  const lp_t* wt = tiled ? p.W_tiled + ((int64_t)blockIdx.x * nd * (2 * NT)) * 512 + lane * 8 : nullptr;
  auto issue = [&](int slot, int i) {
    const int ks = wave + i * SK_WAVES;
    const int k = ks * 64;
    char* st = ring + slot * STAGE;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const lp_t* src = tiled ? wt + ((int64_t)ks * (2 * NT) + t * 2 + h) * 512 : wsrc[t][h] + k;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(st + t * 2048 + h * 1024), 16, 0, 0);
      }
    __builtin_amdgcn_global_load_lds((gptr_t)(asrc + k), (lptr_t)(st + NT * 2048), 16, 0, 0);
  };
  // the first RD stages go out before anything else: neither the weights nor the raw activation rows depend on the statistics
  // (with the fused norm the last wave's last slot carries the row statistics first and is filled after them: two workgroups
  // of 80 KiB are all the LDS a CU has)
  const bool hold_last = NORM && wave == SK_WAVES - 1;
#pragma unroll
  for (int j = 0; j < RD; ++j)
    if (j < n && !(hold_last && j == RD - 1)) issue(j, j);
  // ---- optional fused RMSNorm: row statistics exactly as gemm_skinny_kernel computes them ----
  float rstd = 1.f;
  if (NORM) {
    float* rs_sh = (float*)(smem + (SK_WAVES - 1) * SKR_WAVE_BYTES + (RD - 1) * STAGE);
    // Round 5: the row's loads go out EIGHT AT A TIME and are consumed behind one wait.  With the ring's DMA requests already in
    // flight the compiler guards every ordinary load with `s_waitcnt vmcnt(0)`, so the one-load-per-iteration form paid a memory
    // round trip per 512 elements — eight in series for K = 4096, ~5 us of the ~10 us by which the two norm-fused GEMVs of a layer
    // exceeded their streaming time (profiles/r05_vqa_kernel_stats_*.csv).  Same loads, same summation order (vi ascending, then e):
    // out-of-range slots read a clamped address and contribute +0.
    const int nvec = p.K >> 3;
    for (int row = wave; row < p.M; row += SK_WAVES) {
      const lp_t* xr = p.A + (int64_t)row * p.lda;
      float sum = 0.f;
      for (int v0 = lane; v0 < nvec; v0 += 64 * 8) {
        lpx8 t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int vi = v0 + 64 * j;
          t[j] = *(const lpx8*)(xr + (vi < nvec ? vi : nvec - 1) * 8);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const bool in = v0 + 64 * j < nvec;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float v = in ? lp2f((lp_t)t[j][e]) : 0.f;
            sum += v * v;
          }
        }
      }
      sum = wave_sum(sum);
      if (lane == 0) rs_sh[row] = rsqrtf(sum / (float)p.K + p.norm_eps);
    }
    __syncthreads();
    rstd = rs_sh[arow];
    __syncthreads();
    if (hold_last && RD - 1 < n) issue(RD - 1, RD - 1);
  }
  auto normed = [&](lpx8 x, lpx8 w, float rs) {
    lpx8 y;
#pragma unroll
    for (int e = 0; e < 8; ++e) y[e] = (short)f2lp(lp2f((lp_t)w[e]) * rlp(lp2f((lp_t)x[e]) * rs));
    return y;
  };

  auto consume = [&](int slot) {
    const char* st = ring + slot * STAGE;
    lpx8 wf[2][NT], a[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int t = 0; t < NT; ++t) wf[u][t] = *(const lpx8*)(st + t * 2048 + w_rd[u]);
      a[u] = *(const lpx8*)(st + a_rd[u]);
      if (NORM) a[u] = normed(a[u], *(const lpx8*)(st + n_rd[u]), rstd);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = mfma_16x16x32(wf[u][t], a[u], acc[t]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the slot is re-filled next: its reads must have returned
  };
  int i0 = 0;
  for (; i0 + 2 * RD <= n; i0 += RD) {
    // stages i0 .. i0+RD-1 are in flight; stage i0 + j has landed once at most the RD - 1 younger stages are outstanding
#pragma unroll
    for (int j = 0; j < RD; ++j) {
      static_assert(SOPS * (RD - 1) == 6 || SOPS * (RD - 1) == 5, "add the literal below");
      if constexpr (SOPS * (RD - 1) == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      consume(j);
      issue(j, i0 + RD + j);
    }
  }
  // tail: fewer than 2 RD stages left, the first RD of them (those that exist) are in flight
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int j = 0; j < RD; ++j)
    if (i0 + j < n) consume(j);
#pragma unroll
  for (int j = 0; j < RD; ++j)
    if (i0 + RD + j < n) issue(j, i0 + RD + j);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int j = 0; j < RD; ++j)
    if (i0 + RD + j < n) consume(j);

  // ---- cross-wave reduction in gemm_skinny_kernel's order; wave 0 finishes the (single) row tile ----
  __syncthreads();                                   // every wave is done with its ring
  float (*red)[64][4] = (float (*)[64][4])smem;       // [SK_WAVES][64][4]
  f32x4 s[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (t) __syncthreads();
    *(f32x4*)red[wave][lane] = acc[t];
    __syncthreads();
    if (wave == 0) {
      s[t] = *(const f32x4*)red[0][lane];
#pragma unroll
      for (int w = 1; w < SK_WAVES; ++w) s[t] += *(const f32x4*)red[w][lane];
    }
  }
  if (wave != 0) return;
  const int n_out = (EPI == VSTAR_EPI_SILU_MUL) ? p.N / 2 : p.N;
  if (fr >= p.M) return;
  if (EPI == VSTAR_EPI_SILU_MUL) gemm_epilogue_store<EPI, OUT_F32>(p, fr, n0 / 2 + g * 4, n_out, s[0], s[NT - 1]);
  else gemm_epilogue_store<EPI, OUT_F32>(p, fr, n0 + g * 4, n_out, s[0], s[0]);
}

// one thread per 16-byte chunk of the tile-major image (see GemmParams::W_tiled)
__global__ void skinny_tile_pack_kernel(const lp_t* __restrict__ W, lp_t* __restrict__ Wt, int K, int nt, int64_t n_chunks) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_chunks) return;
  const int lane = (int)(idx & 63);
  const int64_t piece = idx >> 6;
  const int per_step = 2 * nt, nd = K >> 6;
  const int req = (int)(piece % per_step);
  const int64_t r2 = piece / per_step;
  const int ks = (int)(r2 % nd);
  const int64_t wg = r2 / nd;
  const int t = req >> 1, h = req & 1;
  const int row16 = h * 8 + (lane >> 3);
  const int64_t row = wg * 16 * nt + t * 16 + row16;
  const int chunk = (lane & 7) ^ ((row16 >> 1) & 7);          // the source-side swizzle of gemm_skinny_ring_kernel's requests
  *(lpx8*)(Wt + idx * 8) = *(const lpx8*)(W + row * K + ks * 64 + chunk * 8);
}

template <int EPI, bool OUT_F32>
hipError_t launch_skinny_ring(const GemmParams& p0, hipStream_t s) {
  constexpr int NT = (EPI == VSTAR_EPI_SILU_MUL) ? 2 : 1;
  constexpr int lds = SK_WAVES * SKR_WAVE_BYTES;
  GemmParams p = p0;
  if (p.N % (16 * NT)) p.W_tiled = nullptr;                     // whole 16 NT-row tiles only
  const int blocks = (p.N + 16 * NT - 1) / (16 * NT);
  static bool attr_done[2] = {false, false};
  const int nm = p.norm_w ? 1 : 0;
  if (!attr_done[nm]) {
    const void* k = nm ? (const void*)gemm_skinny_ring_kernel<EPI, OUT_F32, true> : (const void*)gemm_skinny_ring_kernel<EPI, OUT_F32, false>;
    hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    attr_done[nm] = true;
  }
  if (nm) hipLaunchKernelGGL((gemm_skinny_ring_kernel<EPI, OUT_F32, true>), dim3(blocks), dim3(SK_WAVES * 64), lds, s, p);
  else hipLaunchKernelGGL((gemm_skinny_ring_kernel<EPI, OUT_F32, false>), dim3(blocks), dim3(SK_WAVES * 64), lds, s, p);
  return hipGetLastError();
}

template <int EPI, bool OUT_F32>
hipError_t launch_skinny(const GemmParams& p, hipStream_t s) {
  constexpr int NT = (EPI == VSTAR_EPI_SILU_MUL) ? 2 : 1;
  const int blocks = (p.N + 16 * NT - 1) / (16 * NT);
  const int mt = (p.M + 15) / 16;
  // M <= 8 (decode steps of up to 8 sequences; 7 with the fused norm): the LDS-ring variant, bit-identical (VSTAR_SKINNY_RING=0: the register-streaming kernel, A/B and tests);
  // the W rows it reads are padded to 256, so whole 16-row tiles exist for every workgroup
  static const bool ring = [] { const char* e = getenv("VSTAR_SKINNY_RING"); return !e || atoi(e) != 0; }();
  if (ring && p.M <= (p.norm_w ? 7 : 8) && p.K >= 512 && p.tile_force != -1) return launch_skinny_ring<EPI, OUT_F32>(p, s);   // tile_force -1: tests
  switch (mt) {
    case 1: hipLaunchKernelGGL((gemm_skinny_kernel<EPI, OUT_F32, 1>), dim3(blocks), dim3(SK_WAVES * 64), 0, s, p); break;
    case 2: hipLaunchKernelGGL((gemm_skinny_kernel<EPI, OUT_F32, 2>), dim3(blocks), dim3(SK_WAVES * 64), 0, s, p); break;
    case 3: hipLaunchKernelGGL((gemm_skinny_kernel<EPI, OUT_F32, 3>), dim3(blocks), dim3(SK_WAVES * 64), 0, s, p); break;
    case 4: hipLaunchKernelGGL((gemm_skinny_kernel<EPI, OUT_F32, 4>), dim3(blocks), dim3(SK_WAVES * 64), 0, s, p); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// ------------------------------------------------ embedding rows ------------------------------------------------
__global__ void embed_rows_kernel(const int32_t* __restrict__ src, const lp_t* __restrict__ table, int vocab,
                                  const lp_t* __restrict__ feats, int64_t n_feat_rows, lp_t* __restrict__ x, int R, int C) {
  const int vec = C >> 3;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)R * vec) return;
  const int r = (int)(idx / vec), v = (int)(idx - (int64_t)r * vec);
  const int sidx = src[r];
  lpx8 o = {0, 0, 0, 0, 0, 0, 0, 0};
  if (sidx >= 0) {
    if (sidx < vocab) o = *(const lpx8*)(table + (int64_t)sidx * C + v * 8);
  } else if (sidx != INT32_MIN) {
    const int64_t f = -(int64_t)sidx - 1;
    if (f < n_feat_rows) o = *(const lpx8*)(feats + f * C + v * 8);
  }
  *(lpx8*)(x + (int64_t)r * C + v * 8) = o;
}

// ------------------------------------------------ RoPE + KV-cache append ------------------------------------------------
// One thread per (row, q|k|v, head, 8-vector of the FIRST half of the head dim); handles d0 and d0 + D/2 together.
__global__ void rope_kv_append_kernel(lp_t* __restrict__ qkv, const lp_t* __restrict__ cos_sin, const int32_t* __restrict__ row_pos,
                                      const int32_t* __restrict__ row_slot, lp_t* __restrict__ kc, lp_t* __restrict__ vc,
                                      int64_t slot_stride, int ctx, int R, int H) {
  constexpr int D = 128, HALF = 64, VPH = HALF / 8;
  const int per_row = 3 * H * VPH;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)R * per_row) return;
  const int row = (int)(idx / per_row);
  int rem = (int)(idx - (int64_t)row * per_row);
  const int which = rem / (H * VPH);
  rem -= which * H * VPH;
  const int h = rem / VPH, d0 = (rem - h * VPH) * 8;
  const int pos = row_pos[row];
  if (pos < 0) return;                                   // padding row of a ragged prefill batch
  lp_t* base = qkv + (int64_t)row * (3 * H * D) + which * (H * D) + h * D;
  lpx8 o1 = *(const lpx8*)(base + d0), o2 = *(const lpx8*)(base + d0 + HALF);
  if (which < 2) {
    const lpx8 c = *(const lpx8*)(cos_sin + (int64_t)pos * D + d0);
    const lpx8 sn = *(const lpx8*)(cos_sin + (int64_t)pos * D + HALF + d0);
    const lpx8 x1 = o1, x2 = o2;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float a = lp2f((lp_t)x1[e]), b = lp2f((lp_t)x2[e]);
      const float cs = lp2f((lp_t)c[e]), si = lp2f((lp_t)sn[e]);
      o1[e] = (short)f2lp(rlp(a * cs) + rlp(-b * si));
      o2[e] = (short)f2lp(rlp(b * cs) + rlp(a * si));
    }
    *(lpx8*)(base + d0) = o1;
    *(lpx8*)(base + d0 + HALF) = o2;
    if (which == 0) return;
  }
  lp_t* dst = (which == 1 ? kc : vc) + (int64_t)row_slot[row] * slot_stride + ((int64_t)h * ctx + pos) * D;
  *(lpx8*)(dst + d0) = o1;
  *(lpx8*)(dst + d0 + HALF) = o2;
}

// ------------------------------------------------ attention over the KV cache ------------------------------------------------
// FUSED (decode steps: every sequence contributes exactly one new row): the workgroup also applies RoPE to its row's q and
// k, appends k and v to the cache and attends to them from LDS — rope_kv_append's work without its launch.
template <bool FUSED>
__global__ __launch_bounds__(256) void cached_attn_kernel(const lp_t* __restrict__ qkv, lp_t* __restrict__ kc, lp_t* __restrict__ vc,
                                                          const int32_t* __restrict__ row_seq, const int32_t* __restrict__ row_pos,
                                                          const int32_t* __restrict__ seq_kv, const int32_t* __restrict__ seq_prefix,
                                                          const int32_t* __restrict__ seq_past, const lp_t* __restrict__ cos_sin,
                                                          lp_t* __restrict__ out, int H, int ctx, int64_t slot_stride,
                                                          float inv_scale) {
  constexpr int D = 128;
  extern __shared__ float dyn[];            // [D] q | [nk] scores/probabilities
  __shared__ float redbuf[8];
  __shared__ float part[16][D];
  __shared__ float own_k[D], own_v[D];
  const int r = blockIdx.x, h = blockIdx.y, tid = threadIdx.x;
  const int pos = row_pos[r];
  if (pos < 0) return;
  const int seq = row_seq[r];
  const int past = seq_past[seq], nk = pos + 1;
  const int nkc = FUSED ? pos : nk;         // keys that come from the cache
  lp_t* kown = kc + (int64_t)seq_kv[seq] * slot_stride + (int64_t)h * ctx * D;
  const lp_t* kpre = kc + (int64_t)seq_prefix[seq] * slot_stride + (int64_t)h * ctx * D;
  lp_t* vown = vc + (int64_t)seq_kv[seq] * slot_stride + (int64_t)h * ctx * D;
  const lp_t* vpre = vc + (int64_t)seq_prefix[seq] * slot_stride + (int64_t)h * ctx * D;
  float* qs = dyn;
  float* sc = dyn + D;
  const lp_t* rowp = qkv + (int64_t)r * (3 * H * D) + h * D;
  if (FUSED) {
    if (tid < 128) {                         // rotate-half RoPE with HF's rounding points: tid 0-63 q pairs, 64-127 k pairs
      const int which = tid >> 6, d = tid & 63;
      const lp_t* base = rowp + which * (H * D);
      const float x1 = lp2f(base[d]), x2 = lp2f(base[d + 64]);
      const float cs = lp2f(cos_sin[(int64_t)pos * D + d]), si = lp2f(cos_sin[(int64_t)pos * D + 64 + d]);
      const lp_t o1 = f2lp(rlp(x1 * cs) + rlp(-x2 * si)), o2 = f2lp(rlp(x2 * cs) + rlp(x1 * si));
      if (which == 0) {
        qs[d] = lp2f(o1);
        qs[d + 64] = lp2f(o2);
      } else {
        own_k[d] = lp2f(o1);
        own_k[d + 64] = lp2f(o2);
        kown[(int64_t)pos * D + d] = o1;
        kown[(int64_t)pos * D + d + 64] = o2;
      }
    } else {
      const int d = tid - 128;
      const lp_t v = rowp[2 * H * D + d];
      own_v[d] = lp2f(v);
      vown[(int64_t)pos * D + d] = v;
    }
  } else if (tid < D) {
    qs[tid] = lp2f(rowp[tid]);
  }
  __syncthreads();
  // ---- scores: 16 lanes per key, 16 key groups, 4 keys per group and pass (64 keys in flight per pass) ----
  const int l16 = tid & 15, grp = tid >> 4;
  float qv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) qv[e] = qs[l16 * 8 + e];
  float mx = -3.0e38f;
  for (int j0 = 0; j0 < nkc; j0 += 64) {
    lpx8 kv8[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = j0 + u * 16 + grp;
      kv8[u] = (lpx8){0, 0, 0, 0, 0, 0, 0, 0};
      if (j < nkc) kv8[u] = *(const lpx8*)((j < past ? kpre : kown) + (int64_t)j * D + l16 * 8);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = j0 + u * 16 + grp;
      float a = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) a += qv[e] * lp2f((lp_t)kv8[u][e]);
      a += __shfl_xor(a, 8, 64);
      a += __shfl_xor(a, 4, 64);
      a += __shfl_xor(a, 2, 64);
      a += __shfl_xor(a, 1, 64);
      if (j < nkc) {
        const float sv = rlp(rlp(a) / inv_scale);    // HF: matmul output in the storage type, then / sqrt(head_dim)
        if (l16 == 0) sc[j] = sv;
        mx = fmaxf(mx, sv);
      }
    }
  }
  if (FUSED && grp == 0) {                     // the row's own key, from LDS
    float a = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) a += qv[e] * own_k[l16 * 8 + e];
    a += __shfl_xor(a, 8, 64);
    a += __shfl_xor(a, 4, 64);
    a += __shfl_xor(a, 2, 64);
    a += __shfl_xor(a, 1, 64);
    const float sv = rlp(rlp(a) / inv_scale);
    if (l16 == 0) sc[pos] = sv;
    mx = fmaxf(mx, sv);
  }
  mx = wave_max(mx);
  if ((tid & 63) == 0) redbuf[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(redbuf[0], redbuf[1]), fmaxf(redbuf[2], redbuf[3]));
  float sum = 0.f;
  for (int j = tid; j < nk; j += 256) {
    const float e = __expf(sc[j] - mx);
    sc[j] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  if ((tid & 63) == 0) redbuf[4 + (tid >> 6)] = sum;
  __syncthreads();
  const float inv = 1.0f / (redbuf[4] + redbuf[5] + redbuf[6] + redbuf[7]);
  // ---- PV: group = keys j == grp (mod 16), lane = 8 output dims (16-byte V loads), 4 keys in flight; probabilities are
  // rounded to the storage type (HF .to(query.dtype)) ----
  float o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = 0.f;
  for (int j0 = grp; j0 < nkc; j0 += 64) {
    lpx8 v8[4];
    float pr[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = j0 + u * 16;
      v8[u] = (lpx8){0, 0, 0, 0, 0, 0, 0, 0};
      pr[u] = 0.f;
      if (j < nkc) {
        v8[u] = *(const lpx8*)((j < past ? vpre : vown) + (int64_t)j * D + l16 * 8);
        pr[u] = rlp(sc[j] * inv);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] += pr[u] * lp2f((lp_t)v8[u][e]);
  }
  if (FUSED && grp == 0) {
    const float pr = rlp(sc[pos] * inv);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] += pr * own_v[l16 * 8 + e];
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) part[grp][l16 * 8 + e] = o[e];
  __syncthreads();
  if (tid < D) {
    float t = 0.f;
#pragma unroll
    for (int g2 = 0; g2 < 16; ++g2) t += part[g2][tid];
    out[(int64_t)r * (H * D) + h * D + tid] = f2lp(t);
  }
}

// ------------------------------------------------ split-KV decode attention ------------------------------------------------
// A decode step of few sequences leaves cached_attn_kernel<true> on R x 32 workgroups of 256 CUs, each pulling the whole K and V
// of its head through one CU (17.8 us per layer at ~700 keys: 15 % of a batch-1 token).  Here the keys of a (row, head) are cut
// into P partitions = P workgroups.  The reference's rounding points need the GLOBAL softmax statistics before a probability is
// rounded to the storage type, so the work is two launches:
//   scores kernel  RoPE(q) (+ RoPE(k), append k / v to the cache: last partition), this partition's scores -> workspace, its
//                  local max and sum of exponentials
//   pv kernel      global max / sum from the P partials (fixed order), probabilities rounded like HF, partial P.V; the LAST
//                  partition to finish (ticket) adds the partials in partition order and writes the row — no spinning, no
//                  co-residency requirement, deterministic summation order.
// Same arithmetic per key as cached_attn_kernel; only the order of the fp32 sums differs.
constexpr int SPLIT_P = 8;

__device__ __forceinline__ void split_range(int nkc, int p, int* lo, int* hi) {
  const int span = (((nkc + SPLIT_P - 1) / SPLIT_P) + 63) & ~63;
  *lo = p * span < nkc ? p * span : nkc;
  *hi = *lo + span < nkc ? *lo + span : nkc;
}

__global__ __launch_bounds__(256) void cached_attn_split_scores_kernel(
    const lp_t* __restrict__ qkv, lp_t* __restrict__ kc, lp_t* __restrict__ vc, const int32_t* __restrict__ row_seq,
    const int32_t* __restrict__ row_pos, const int32_t* __restrict__ seq_kv, const int32_t* __restrict__ seq_prefix,
    const int32_t* __restrict__ seq_past, const lp_t* __restrict__ cos_sin, float* __restrict__ ws_scores, float* __restrict__ ws_stats,
    int H, int ctx, int64_t slot_stride, float inv_scale) {
  constexpr int D = 128;
  extern __shared__ float dyn[];            // this partition's scores (span + 1)
  __shared__ float qs[D], own_k[D], redbuf[8];
  const int r = blockIdx.x, h = blockIdx.y, p = blockIdx.z, tid = threadIdx.x;
  const int pos = row_pos[r];
  if (pos < 0) return;
  const int seq = row_seq[r];
  const int past = seq_past[seq], nkc = pos;
  lp_t* kown = kc + (int64_t)seq_kv[seq] * slot_stride + (int64_t)h * ctx * D;
  const lp_t* kpre = kc + (int64_t)seq_prefix[seq] * slot_stride + (int64_t)h * ctx * D;
  lp_t* vown = vc + (int64_t)seq_kv[seq] * slot_stride + (int64_t)h * ctx * D;
  const lp_t* rowp = qkv + (int64_t)r * (3 * H * D) + h * D;
  const bool last = p == SPLIT_P - 1;
  if (tid < 128) {                           // rotate-half RoPE with HF's rounding points: tid 0-63 q pairs, 64-127 k pairs
    const int which = tid >> 6, d = tid & 63;
    if (which == 0 || last) {
      const lp_t* base = rowp + which * (H * D);
      const float x1 = lp2f(base[d]), x2 = lp2f(base[d + 64]);
      const float cs = lp2f(cos_sin[(int64_t)pos * D + d]), si = lp2f(cos_sin[(int64_t)pos * D + 64 + d]);
      const lp_t o1 = f2lp(rlp(x1 * cs) + rlp(-x2 * si)), o2 = f2lp(rlp(x2 * cs) + rlp(x1 * si));
      if (which == 0) {
        qs[d] = lp2f(o1);
        qs[d + 64] = lp2f(o2);
      } else {
        own_k[d] = lp2f(o1);
        own_k[d + 64] = lp2f(o2);
        kown[(int64_t)pos * D + d] = o1;
        kown[(int64_t)pos * D + d + 64] = o2;
      }
    }
  } else if (last) {
    const int d = tid - 128;
    vown[(int64_t)pos * D + d] = rowp[2 * H * D + d];
  }
  __syncthreads();
  int j_lo, j_hi;
  split_range(nkc, p, &j_lo, &j_hi);
  float* gsc = ws_scores + ((int64_t)r * H + h) * ctx;
  const int l16 = tid & 15, grp = tid >> 4;
  float qv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) qv[e] = qs[l16 * 8 + e];
  float mx = -3.0e38f;
  for (int j0 = j_lo; j0 < j_hi; j0 += 64) {
    lpx8 kv8[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = j0 + u * 16 + grp;
      kv8[u] = (lpx8){0, 0, 0, 0, 0, 0, 0, 0};
      if (j < j_hi) kv8[u] = *(const lpx8*)((j < past ? kpre : kown) + (int64_t)j * D + l16 * 8);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = j0 + u * 16 + grp;
      float a = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) a += qv[e] * lp2f((lp_t)kv8[u][e]);
      a += __shfl_xor(a, 8, 64);
      a += __shfl_xor(a, 4, 64);
      a += __shfl_xor(a, 2, 64);
      a += __shfl_xor(a, 1, 64);
      if (j < j_hi) {
        const float sv = rlp(rlp(a) / inv_scale);    // HF: matmul output in the storage type, then / sqrt(head_dim)
        if (l16 == 0) { dyn[j - j_lo] = sv; gsc[j] = sv; }
        mx = fmaxf(mx, sv);
      }
    }
  }
  int n_loc = j_hi - j_lo;
  if (last) {                                  // the row's own key, from LDS
    if (grp == 0) {
      float a = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) a += qv[e] * own_k[l16 * 8 + e];
      a += __shfl_xor(a, 8, 64);
      a += __shfl_xor(a, 4, 64);
      a += __shfl_xor(a, 2, 64);
      a += __shfl_xor(a, 1, 64);
      const float sv = rlp(rlp(a) / inv_scale);
      if (l16 == 0) { dyn[n_loc] = sv; gsc[pos] = sv; }
      mx = fmaxf(mx, sv);
    }
    n_loc += 1;
  }
  mx = wave_max(mx);
  if ((tid & 63) == 0) redbuf[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(redbuf[0], redbuf[1]), fmaxf(redbuf[2], redbuf[3]));
  float sum = 0.f;
  for (int j = tid; j < n_loc; j += 256) sum += __expf(dyn[j] - mx);
  sum = wave_sum(sum);
  if ((tid & 63) == 0) redbuf[4 + (tid >> 6)] = sum;
  __syncthreads();
  if (tid == 0) {
    float* st = ws_stats + (((int64_t)r * H + h) * SPLIT_P + p) * 2;
    st[0] = mx;
    st[1] = (redbuf[4] + redbuf[5]) + (redbuf[6] + redbuf[7]);
  }
}

__global__ __launch_bounds__(256) void cached_attn_split_pv_kernel(
    const lp_t* __restrict__ vc, const int32_t* __restrict__ row_seq, const int32_t* __restrict__ row_pos,
    const int32_t* __restrict__ seq_kv, const int32_t* __restrict__ seq_prefix, const int32_t* __restrict__ seq_past,
    const float* __restrict__ ws_scores, const float* __restrict__ ws_stats, float* __restrict__ ws_opart, int* __restrict__ ws_cnt,
    lp_t* __restrict__ out, int H, int ctx, int64_t slot_stride) {
  constexpr int D = 128;
  __shared__ float part[16][D];
  __shared__ int ticket;
  const int r = blockIdx.x, h = blockIdx.y, p = blockIdx.z, tid = threadIdx.x;
  const int pos = row_pos[r];
  if (pos < 0) return;
  const int seq = row_seq[r];
  const int past = seq_past[seq], nkc = pos;
  const lp_t* vown = vc + (int64_t)seq_kv[seq] * slot_stride + (int64_t)h * ctx * D;
  const lp_t* vpre = vc + (int64_t)seq_prefix[seq] * slot_stride + (int64_t)h * ctx * D;
  const float* st = ws_stats + ((int64_t)r * H + h) * SPLIT_P * 2;
  float M = -3.0e38f;
#pragma unroll
  for (int q = 0; q < SPLIT_P; ++q) M = fmaxf(M, st[2 * q]);
  float L = 0.f;
#pragma unroll
  for (int q = 0; q < SPLIT_P; ++q) L += st[2 * q + 1] * __expf(st[2 * q] - M);
  const float inv = 1.0f / L;
  int j_lo, j_hi;
  split_range(nkc, p, &j_lo, &j_hi);
  if (p == SPLIT_P - 1) j_hi = (j_hi == nkc) ? nkc + 1 : j_hi;       // + the row's own key / value (appended by the scores kernel)
  const float* gsc = ws_scores + ((int64_t)r * H + h) * ctx;
  const int l16 = tid & 15, grp = tid >> 4;
  float o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = 0.f;
  for (int j0 = j_lo + grp; j0 < j_hi; j0 += 64) {
    lpx8 v8[4];
    float pr[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = j0 + u * 16;
      v8[u] = (lpx8){0, 0, 0, 0, 0, 0, 0, 0};
      pr[u] = 0.f;
      if (j < j_hi) {
        v8[u] = *(const lpx8*)((j < past ? vpre : vown) + (int64_t)j * D + l16 * 8);
        pr[u] = rlp(__expf(gsc[j] - M) * inv);     // probabilities rounded to the storage type (HF .to(query.dtype))
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] += pr[u] * lp2f((lp_t)v8[u][e]);
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) part[grp][l16 * 8 + e] = o[e];
  __syncthreads();
  float* op = ws_opart + ((int64_t)r * H + h) * SPLIT_P * D;
  // Cross-workgroup hand-over WITHOUT fences: a __threadfence() here is a whole-L2 write-back + invalidate per workgroup on this
  // chip (first version of this kernel: 4.18 ms / token against 3.79 unsplit).  The partials are written with agent-scope atomic
  // stores (write-through to the coherence point, nothing left dirty in this XCD's L2), the wave waits until they are performed
  // (vmcnt(0)), then takes its ticket with an agent-scope atomic; the last workgroup reads the partials with agent-scope atomic
  // loads (never served from a stale line of its own L2).  Per-location coherence of atomics + completion of the stores before
  // the ticket is exactly the ordering needed.
  if (tid < D) {
    float t = 0.f;
#pragma unroll
    for (int g2 = 0; g2 < 16; ++g2) t += part[g2][tid];
    __hip_atomic_store(&op[p * D + tid], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) ticket = __hip_atomic_fetch_add(&ws_cnt[r * H + h], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (ticket != SPLIT_P - 1) return;
  if (tid < D) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < SPLIT_P; ++q) t += __hip_atomic_load(&op[q * D + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    out[(int64_t)r * (H * D) + h * D + tid] = f2lp(t);
  }
  if (tid == 0) __hip_atomic_store(&ws_cnt[r * H + h], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
}

// ------------------------------------------------ Perceiver attention ------------------------------------------------
// q [n*L, H*DH]; kv [n*NK, 2*H*DH] = k | v; out [n*L, H*DH].  Rounding points of the fp16 reference: q*scale, sim, sim-amax,
// softmax output and the attn·v product are each materialised in the storage type.
template <int DH>
__global__ __launch_bounds__(256) void perceiver_attn_kernel(const lp_t* __restrict__ q, const lp_t* __restrict__ kv,
                                                             lp_t* __restrict__ out, int n, int L, int NK, int H, float scale) {
  constexpr int MAXK_PER_LANE = 8;     // up to 512 keys
  __shared__ float qsh[4][DH];
  __shared__ float psh[4][64 * MAXK_PER_LANE];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t wid = (int64_t)blockIdx.x * 4 + w;
  if (wid >= (int64_t)n * H * L) return;
  const int qi = (int)(wid % L), h = (int)((wid / L) % H), b = (int)(wid / ((int64_t)L * H));
  const int C = H * DH;
  const lp_t* qp = q + ((int64_t)b * L + qi) * C + h * DH;
  for (int d = lane; d < DH; d += 64) qsh[w][d] = rlp(lp2f(qp[d]) * scale);
  __builtin_amdgcn_wave_barrier();
  float sc[MAXK_PER_LANE];
  float mx = -3.0e38f;
#pragma unroll
  for (int i = 0; i < MAXK_PER_LANE; ++i) {
    const int key = i * 64 + lane;
    float s = -3.0e38f;
    if (key < NK) {
      const lp_t* kp = kv + ((int64_t)b * NK + key) * (2 * C) + h * DH;
      float a = 0.f;
#pragma unroll
      for (int d8 = 0; d8 < DH / 8; ++d8) {
        const lpx8 k8 = *(const lpx8*)(kp + d8 * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) a += qsh[w][d8 * 8 + e] * lp2f((lp_t)k8[e]);
      }
      s = rlp(a);
    }
    sc[i] = s;
    mx = fmaxf(mx, s);
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < MAXK_PER_LANE; ++i) {
    const int key = i * 64 + lane;
    if (key < NK) {
      const float e = __expf(rlp(sc[i] - mx));
      sc[i] = e;
      sum += e;
    }
  }
  sum = wave_sum(sum);
  const float inv = 1.0f / sum;
#pragma unroll
  for (int i = 0; i < MAXK_PER_LANE; ++i) {
    const int key = i * 64 + lane;
    if (key < NK) psh[w][key] = rlp(sc[i] * inv);
  }
  __builtin_amdgcn_wave_barrier();
  if (lane * 2 < DH) {
    const lp_t* vp = kv + (int64_t)b * NK * (2 * C) + C + h * DH + lane * 2;
    float o0 = 0.f, o1 = 0.f;
    for (int key = 0; key < NK; ++key) {
      const uint32_t v2 = *(const uint32_t*)(vp + (int64_t)key * (2 * C));
      const float pr = psh[w][key];
      o0 += pr * lp2f((lp_t)(v2 & 0xffff));
      o1 += pr * lp2f((lp_t)(v2 >> 16));
    }
    lp_t* op = out + ((int64_t)b * L + qi) * C + h * DH + lane * 2;
    op[0] = f2lp(o0);
    op[1] = f2lp(o1);
  }
}

__global__ void argmax_rows_lp_kernel(const lp_t* __restrict__ x, int cols, int64_t ld, int32_t* __restrict__ out) {
  __shared__ float bv[4];
  __shared__ int bi[4];
  const int row = blockIdx.x;
  const lp_t* p = x + (int64_t)row * ld;
  float best = -3.0e38f;
  int idx = 0;
  for (int c = threadIdx.x; c < cols; c += blockDim.x) {
    const float v = lp2f(p[c]);
    if (v > best) { best = v; idx = c; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(idx, o, 64);
    if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
  }
  if ((threadIdx.x & 63) == 0) { bv[threadIdx.x >> 6] = best; bi[threadIdx.x >> 6] = idx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w)
      if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
    out[row] = idx;
  }
}

}  // namespace

bool gemm_skinny_eligible(const GemmParams& p) {
  return p.M > 0 && p.M <= 64 && p.a_group <= 0 && p.c_group <= 0 && p.K % 64 == 0 && (p.lda % 8) == 0;
}

hipError_t skinny_pack_tiles(const lp_t* W, lp_t* Wt, int n_rows, int K, int nt, hipStream_t s) {
  if (n_rows <= 0 || K <= 0 || K % 64 || (nt != 1 && nt != 2) || n_rows % (16 * nt)) return hipErrorInvalidValue;
  const int64_t n_chunks = (int64_t)n_rows * K / 8;
  hipLaunchKernelGGL(skinny_tile_pack_kernel, dim3((unsigned)((n_chunks + 255) / 256)), dim3(256), 0, s, W, Wt, K, nt, n_chunks);
  return hipGetLastError();
}

hipError_t gemm_skinny_lp(const GemmParams& p, int epilogue, bool out_f32, hipStream_t s) {
  if (!gemm_skinny_eligible(p)) return hipErrorInvalidValue;
#define SK_CASE(E)                                                                   \
  case E:                                                                            \
    return out_f32 ? launch_skinny<E, true>(p, s) : launch_skinny<E, false>(p, s);
  switch (epilogue) {
    SK_CASE(VSTAR_EPI_NONE)
    SK_CASE(VSTAR_EPI_QUICK_GELU)
    SK_CASE(VSTAR_EPI_GELU)
    SK_CASE(VSTAR_EPI_RELU)
    SK_CASE(VSTAR_EPI_SILU_MUL)
  }
#undef SK_CASE
  return hipErrorInvalidValue;
}

hipError_t embed_rows(const int32_t* src, const lp_t* table, int vocab, const lp_t* feats, int64_t n_feat_rows, lp_t* x, int R,
                      int C, hipStream_t s) {
  if (R <= 0) return hipSuccess;
  const int64_t n = (int64_t)R * (C / 8);
  hipLaunchKernelGGL(embed_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, table, vocab, feats,
                     n_feat_rows, x, R, C);
  return hipGetLastError();
}

hipError_t rope_kv_append(lp_t* qkv, const lp_t* cos_sin, const int32_t* row_pos, const int32_t* row_slot, lp_t* kc, lp_t* vc,
                          int64_t slot_stride, int ctx, int R, int H, hipStream_t s) {
  if (R <= 0) return hipSuccess;
  const int64_t n = (int64_t)R * 3 * H * 8;
  hipLaunchKernelGGL(rope_kv_append_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, qkv, cos_sin, row_pos, row_slot,
                     kc, vc, slot_stride, ctx, R, H);
  return hipGetLastError();
}

size_t cached_attention_split_ws_bytes(int max_rows, int H, int ctx) {
  const size_t rh = (size_t)max_rows * H;
  return rh * ((size_t)ctx * 4 + SPLIT_P * 2 * 4 + SPLIT_P * 128 * 4 + 4) + 256;
}

hipError_t cached_attention(const lp_t* qkv, lp_t* kc, lp_t* vc, const int32_t* row_seq, const int32_t* row_pos,
                            const int32_t* seq_kv, const int32_t* seq_prefix, const int32_t* seq_past, const lp_t* fused_cos_sin,
                            lp_t* out, int R, int H, int ctx, int64_t slot_stride, int max_keys, hipStream_t s, void* split_ws,
                            int split_max_rows) {
  if (R <= 0) return hipSuccess;
  static const bool split_on = [] { const char* v = getenv("VSTAR_DECODE_SPLIT_KV"); return !v || atoi(v) != 0; }();
  // decode steps of few sequences: P partitions per (row, head) so that the K / V streams of a head run on 8 CUs instead of one
  if (split_on && fused_cos_sin && split_ws && R <= split_max_rows && R * H <= 128 && max_keys >= 256) {
    const size_t rh = (size_t)split_max_rows * H;
    float* ws_scores = (float*)split_ws;
    float* ws_stats = ws_scores + rh * ctx;
    float* ws_opart = ws_stats + rh * SPLIT_P * 2;
    int* ws_cnt = (int*)(ws_opart + rh * SPLIT_P * 128);
    const int span = ((((max_keys + SPLIT_P - 1) / SPLIT_P) + 63) & ~63) + 1;
    hipLaunchKernelGGL(cached_attn_split_scores_kernel, dim3(R, H, SPLIT_P), dim3(256), (size_t)span * sizeof(float), s, qkv, kc, vc,
                       row_seq, row_pos, seq_kv, seq_prefix, seq_past, fused_cos_sin, ws_scores, ws_stats, H, ctx, slot_stride,
                       sqrtf(128.0f));
    hipLaunchKernelGGL(cached_attn_split_pv_kernel, dim3(R, H, SPLIT_P), dim3(256), 0, s, vc, row_seq, row_pos, seq_kv, seq_prefix,
                       seq_past, ws_scores, ws_stats, ws_opart, ws_cnt, out, H, ctx, slot_stride);
    return hipGetLastError();
  }
  const size_t lds = (size_t)(128 + max_keys) * sizeof(float);
  if (lds > 40 * 1024) return hipErrorInvalidValue;
  if (fused_cos_sin)
    hipLaunchKernelGGL(cached_attn_kernel<true>, dim3(R, H), dim3(256), lds, s, qkv, kc, vc, row_seq, row_pos, seq_kv, seq_prefix,
                       seq_past, fused_cos_sin, out, H, ctx, slot_stride, sqrtf(128.0f));
  else
    hipLaunchKernelGGL(cached_attn_kernel<false>, dim3(R, H), dim3(256), lds, s, qkv, kc, vc, row_seq, row_pos, seq_kv, seq_prefix,
                       seq_past, fused_cos_sin, out, H, ctx, slot_stride, sqrtf(128.0f));
  return hipGetLastError();
}

hipError_t perceiver_attention(const lp_t* q, const lp_t* kv, lp_t* out, int n, int L, int NK, int H, int DH, hipStream_t s) {
  if (DH != 96 && DH != 64 && DH != 32) return hipErrorInvalidValue;
  if (NK > 512 || n <= 0) return hipErrorInvalidValue;
  const int64_t waves = (int64_t)n * H * L;
  const dim3 grid((unsigned)((waves + 3) / 4));
  const float scale = 1.0f / sqrtf((float)DH);
  if (DH == 96) hipLaunchKernelGGL(perceiver_attn_kernel<96>, grid, dim3(256), 0, s, q, kv, out, n, L, NK, H, scale);
  else if (DH == 64) hipLaunchKernelGGL(perceiver_attn_kernel<64>, grid, dim3(256), 0, s, q, kv, out, n, L, NK, H, scale);
  else hipLaunchKernelGGL(perceiver_attn_kernel<32>, grid, dim3(256), 0, s, q, kv, out, n, L, NK, H, scale);
  return hipGetLastError();
}

hipError_t argmax_rows_lp(const lp_t* x, int rows, int cols, int64_t ld, int32_t* out, hipStream_t s) {
  if (rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(argmax_rows_lp_kernel, dim3(rows), dim3(256), 0, s, x, cols, ld, out);
  return hipGetLastError();
}

}  // namespace VS_NS
