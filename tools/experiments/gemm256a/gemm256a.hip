// gemm256a.hip — EXPERIMENT (round 3): the 256 x 256 x 64 bf16 GEMM tile with FOUR waves (one per SIMD), each owning a 128 x 128
// output in 256 AGPR accumulators, and a hand-scheduled K loop (tools/gen_gemm256a_asm.py -> gemm256a_loop.inc): the LDS fragment
// reads and the LDS-DMA pieces are interleaved between the MFMAs of a wave instead of sitting in a separate load phase of a
// partner wave, the fragment traffic through LDS is one third lower than in the 8-wave kernel (128 x 128 instead of 128 x 64 per
// wave), and there is one barrier per K-tile instead of four rendezvous.  hipcc cannot express this (256 accumulators + two sets
// of operand fragments exceed what it will keep in registers without shuttling v_accvgpr moves), hence the generated assembly.
//
// Selected with VSTAR_GEMM256A=1 (A/B against gemm256.hip); shapes: M >= 1024, K % 128 == 0, identity row maps, bf16 output,
// epilogues NONE (+bias, +residual) only — the fused RoPE / folded-norm / SiLU epilogues stay on gemm256.hip.
// Same accumulation order as every other kernel (k ascending, 32 per MFMA): bit-identical results.
#include "common.hpp"
#include "kernels.hpp"
#include "gemm_epilogue.hpp"
#include "gemm256a_loop.inc"

namespace VS_NS {

#ifndef VSTAR_LP_F16

namespace {

constexpr int BM = 256, BN = 256;
constexpr int LDS_TOTAL = 2 * 65536;

template <int IDX>
__device__ __forceinline__ float agpr_read() {
  float x;
  asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(x) : "n"(IDX));
  return x;
}

template <int M, int N>
__device__ __forceinline__ f32x4 acc_frag() {
  constexpr int B = (M * 8 + N) * 4;
  return (f32x4){agpr_read<B>(), agpr_read<B + 1>(), agpr_read<B + 2>(), agpr_read<B + 3>()};
}

template <int M>
__device__ __forceinline__ void store_row_frags(const GemmParams& p, int row, int col0, int n_out) {
  if (row >= p.M) return;
  const int64_t crow = row;
  gemm_epilogue_store<VSTAR_EPI_NONE, false>(p, crow, col0 + 0 * 16, n_out, acc_frag<M, 0>(), acc_frag<M, 0>());
  gemm_epilogue_store<VSTAR_EPI_NONE, false>(p, crow, col0 + 1 * 16, n_out, acc_frag<M, 1>(), acc_frag<M, 1>());
  gemm_epilogue_store<VSTAR_EPI_NONE, false>(p, crow, col0 + 2 * 16, n_out, acc_frag<M, 2>(), acc_frag<M, 2>());
  gemm_epilogue_store<VSTAR_EPI_NONE, false>(p, crow, col0 + 3 * 16, n_out, acc_frag<M, 3>(), acc_frag<M, 3>());
  gemm_epilogue_store<VSTAR_EPI_NONE, false>(p, crow, col0 + 4 * 16, n_out, acc_frag<M, 4>(), acc_frag<M, 4>());
  gemm_epilogue_store<VSTAR_EPI_NONE, false>(p, crow, col0 + 5 * 16, n_out, acc_frag<M, 5>(), acc_frag<M, 5>());
  gemm_epilogue_store<VSTAR_EPI_NONE, false>(p, crow, col0 + 6 * 16, n_out, acc_frag<M, 6>(), acc_frag<M, 6>());
  gemm_epilogue_store<VSTAR_EPI_NONE, false>(p, crow, col0 + 7 * 16, n_out, acc_frag<M, 7>(), acc_frag<M, 7>());
}

__global__ __launch_bounds__(256, 1) void gemm256a_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // ---- tile id: XCD-aware bijective remap + GROUP_M = 4 ordering, as in gemm256.hip ----
  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
  const int nwg = tiles_m * tiles_n;
  int t;
  {
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  constexpr int GROUP_M = 4;
  const int in_group = GROUP_M * tiles_n;
  const int grp = t / in_group;
  const int first_m = grp * GROUP_M;
  const int gsz = min(tiles_m - first_m, GROUP_M);
  const int rem = t - grp * in_group;
  const int m0 = (first_m + rem % gsz) * BM, n0 = (rem / gsz) * BN;

  // ---- DMA sources: wave w moves the 8-row pieces g = 8 w + i of the A tile and of the W tile ----
  const int st_r = lane >> 3, st_c = lane & 7;
  uint32_t va[8], vw[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = (wave * 8 + i) * 8 + st_r;
    const int cg = st_c ^ ((row >> 1) & 7);
    int ar = m0 + row;
    ar = ar < p.M ? ar : p.M - 1;
    va[i] = (uint32_t)((int64_t)ar * p.lda * 2 + cg * 16);
    vw[i] = (uint32_t)((int64_t)(n0 + row) * p.K * 2 + cg * 16);
  }
  // ---- fragment read addresses (buffer 0): row-major 128-B rows, chunk ^ ((row >> 1) & 7) ----
  const int wr = wave >> 1, wc = wave & 1;
  const int fr = lane & 15, fq = lane >> 4;
  const int swz = (fr >> 1) & 7;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  uint32_t rd[4];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int ch = ((kk * 4 + fq) ^ swz) * 16;
    rd[kk] = lds0 + (wr * 128 + fr) * 128 + ch;
    rd[2 + kk] = lds0 + 32768 + (wc * 128 + fr) * 128 + ch;
  }
  // ---- L2 prefetch duty of this CU (see tools/gen_gemm256a_asm.py::prefetch_ops): the 32 CUs of an XCD work on a patch of
  // GROUP_M x 8 tiles; of the A row-tile's 256 lines per K-tile this CU touches the 32 of its patch column, of the W column-tile's
  // 256 lines the 64 of its patch row — early, as one dword load per wave and operand ----
  const int pi = rem % gsz, pj = ((blockIdx.x >> 3) >> 2) & 7;
  int prow_a = m0 + pj * 32 + wave * 8 + (lane & 7);
  prow_a = prow_a < p.M ? prow_a : p.M - 1;
  const int prow_w = n0 + (pi & 3) * 64 + wave * 16 + (lane & 15);
  const uint32_t kb_last = (uint32_t)(p.K - 64) * 2;
  const uint32_t pfa0 = (uint32_t)((int64_t)prow_a * p.lda * 2), pfw0 = (uint32_t)((int64_t)prow_w * p.K * 2);
  const uint32_t pfamax = pfa0 + kb_last, pfwmax = pfw0 + kb_last;
  #ifndef G256A_PF_LEAD
#define G256A_PF_LEAD 3
#endif
  const uint32_t pf_lead = G256A_PF_LEAD * 128;        // K-tile T + 3 while K-tile T is being computed
  const uint32_t pfa = pfa0 + (pf_lead < kb_last ? pf_lead : kb_last), pfw = pfw0 + (pf_lead < kb_last ? pf_lead : kb_last);
  const uint32_t ldsw = __builtin_amdgcn_readfirstlane(lds0 + wave * 8192);
  uint32_t cnt = (uint32_t)(p.K / 128 - 1);          // two K-tiles per loop iteration, the last pair is peeled
  const lp_t* abase = p.A;
  const lp_t* wbase = p.W;

  asm volatile(GEMM256A_LOOP_ASM
               : [cnt] "+s"(cnt)
               : [abase] "s"(abase), [wbase] "s"(wbase), [ldsw] "s"(ldsw), [rd0] "v"(rd[0]), [rd1] "v"(rd[1]), [rd2] "v"(rd[2]),
                 [rd3] "v"(rd[3]), [va0] "v"(va[0]), [va1] "v"(va[1]), [va2] "v"(va[2]), [va3] "v"(va[3]), [va4] "v"(va[4]),
                 [va5] "v"(va[5]), [va6] "v"(va[6]), [va7] "v"(va[7]), [vw0] "v"(vw[0]), [vw1] "v"(vw[1]), [vw2] "v"(vw[2]),
                 [vw3] "v"(vw[3]), [vw4] "v"(vw[4]), [vw5] "v"(vw[5]), [vw6] "v"(vw[6]), [vw7] "v"(vw[7]), [pfa] "v"(pfa),
                 [pfw] "v"(pfw), [pfamax] "v"(pfamax), [pfwmax] "v"(pfwmax)
               : GEMM256A_CLOBBERS);

  // ---- epilogue (stage 1 of the experiment): accumulator-layout stores, bias / residual through gemm_epilogue_store ----
  const int n_out = p.N;
  const int rbase = m0 + wr * 128 + fr, cbase = n0 + wc * 128 + fq * 4;
  store_row_frags<0>(p, rbase + 0 * 16, cbase, n_out);
  store_row_frags<1>(p, rbase + 1 * 16, cbase, n_out);
  store_row_frags<2>(p, rbase + 2 * 16, cbase, n_out);
  store_row_frags<3>(p, rbase + 3 * 16, cbase, n_out);
  store_row_frags<4>(p, rbase + 4 * 16, cbase, n_out);
  store_row_frags<5>(p, rbase + 5 * 16, cbase, n_out);
  store_row_frags<6>(p, rbase + 6 * 16, cbase, n_out);
  store_row_frags<7>(p, rbase + 7 * 16, cbase, n_out);
}

}  // namespace

bool gemm256a_eligible(const GemmParams& p, int epilogue, bool out_f32) {
  if (epilogue != VSTAR_EPI_NONE || out_f32) return false;
  if (p.a_scale || p.rope_cs || p.row_scale || p.sumsq_out || p.norm_w || p.a_group > 0 || p.c_group > 0 || p.debug_flags) return false;
  if (p.K % 128 != 0 || p.K < 128 || p.M < 1024 || p.N < 256) return false;
  const int64_t npad = ((int64_t)p.N + BN - 1) / BN * BN;
  if ((int64_t)p.M * p.lda * 2 >= (1ll << 32) || npad * p.K * 2 >= (1ll << 32)) return false;
  return true;
}

hipError_t gemm256a_lp(const GemmParams& p, hipStream_t s) {
  if (gemm_plan_only()) return hipSuccess;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm256a_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  hipLaunchKernelGGL(gemm256a_kernel, dim3(tiles), dim3(256), LDS_TOTAL, s, p);
  return hipGetLastError();
}

#else   // fp16 instantiation: the experiment is bf16 only

bool gemm256a_eligible(const GemmParams&, int, bool) { return false; }
hipError_t gemm256a_lp(const GemmParams&, hipStream_t) { return hipErrorInvalidValue; }

#endif

}  // namespace VS_NS
