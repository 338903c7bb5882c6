// attention.hip — fused softmax(Q K^T * scale [+ causal mask]) V for gfx950, plus RoPE / V-transpose prep.
//
// Replaces HF CLIPAttention / OwlViTAttention (non-causal, 64-d heads; reached from clip_encoder.py:53-57 and
// owlvit.py:121-126) and HF LlamaAttention (causal, 128-d heads, rotate-half RoPE; llava_llama.py:93-102),
// and the SAM two-way transformer's Attention (segment_anything/modeling/transformer.py:185-242).
//
// attn_forward (attn2_kernel): one wave64 owns 32 query rows; v_mfma_f32_32x32x16_bf16 for both products; the 64-key K tile
// and V^T tile are DMA'd into a double-buffered LDS ring shared by the block's 4 waves.
//   S^T = K Q^T  (K tile as the A operand, Q^T as B): each lane ends up with ONE query column (lane&31) and 16 of
//   the tile's 32 keys, so the online-softmax statistics are per-lane scalars (one shuffle with lane^32).
//   O^T = V^T P^T: P^T is fed straight from the S^T accumulator registers (the contraction order over keys is
//   permuted identically on the V^T side), so no cross-lane movement or LDS round trip for P.
//   V tiles are DMA'd untransposed ([key][d], as they lie in the qkv buffer) and transposed on the way out of LDS by
//   ds_read_b64_tr_b16 (two transpose reads per PV operand fragment): no V^T pass, no V^T buffer.
#include "common.hpp"
#include "kernels.hpp"
#include "mx.hpp"
#include <cstdlib>
#include <type_traits>

#ifndef ATTN_EARLY_V
#define ATTN_EARLY_V(D) ((D) == 128)
#endif

namespace VS_NS {

namespace {

// ---------------- RoPE (in place on q and k of the fused qkv buffer) ----------------
// HF rotate-half convention with the reference's bf16 rounding points: each product and the sum are bf16.
__global__ void rope_kernel(lp_t* __restrict__ qkv, const lp_t* __restrict__ cos_sin, int rows, int S, int H, int D, int grp_R0,
                            int grp_Lc) {
  const int half = D >> 1;
  const int vec_per_head = half >> 3;                 // 8-element vectors in the first half
  const int per_row = 2 * H * vec_per_head;           // q and k
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)rows * per_row) return;
  const int row = (int)(idx / per_row);
  int rem = (int)(idx - (int64_t)row * per_row);
  const int which = rem / (H * vec_per_head);         // 0 = q, 1 = k
  rem -= which * H * vec_per_head;
  const int h = rem / vec_per_head;
  const int d0 = (rem - h * vec_per_head) * 8;
  int pos = row % S;
  if (grp_R0 > 0 && pos >= grp_R0) pos = grp_Lc + ((pos - grp_R0) & 31);      // grouped sequences: 32-row suffix blocks restart at grp_Lc
  lp_t* base = qkv + (int64_t)row * (3 * H * D) + which * (H * D) + h * D;
  const lpx8 x1 = *(const lpx8*)(base + d0);
  const lpx8 x2 = *(const lpx8*)(base + d0 + half);
  const lpx8 c = *(const lpx8*)(cos_sin + (int64_t)pos * D + d0);
  const lpx8 sn = *(const lpx8*)(cos_sin + (int64_t)pos * D + half + d0);
  lpx8 o1, o2;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float a = lp2f((lp_t)x1[e]), b = lp2f((lp_t)x2[e]);
    const float cs = lp2f((lp_t)c[e]), si = lp2f((lp_t)sn[e]);
    o1[e] = (short)f2lp(rlp(a * cs) + rlp(-b * si));
    o2[e] = (short)f2lp(rlp(b * cs) + rlp(a * si));
  }
  *(lpx8*)(base + d0) = o1;
  *(lpx8*)(base + d0 + half) = o2;
}

// ---------------- flash attention forward, v2: K / V^T tiles staged through LDS ----------------
// Same per-wave math as attn_kernel, but the 64-key K tile [64][D] and V^T tile [D][64] are DMA'd once per block
// (global_load_lds, whole 128/256-byte lines) into a double-buffered LDS ring and shared by the block's 4 waves, instead of
// every wave issuing fragment-shaped global loads (32 cache lines per instruction).  XOR chunk swizzles (applied on the
// DMA source address and on the read address) keep the ds_read_b128 K-fragment reads conflict-free and the ds_read_b64
// V^T reads at most 2-way.  One counted vmcnt + two barriers per 64-key tile; next tile's DMA is in flight during compute.
// MXO (W8A8 mode, mx.hpp): the output leaves block-scaled — fp8 bytes to (uint8_t*)out and one E8M0 byte per (row, 32 d) to `mxs`
// (tile-major for the o_proj GEMM: a head = one K-tile, m128 = rows / 128) — quantised from the 16-bit value the plain kernel stores.
template <int D, bool CAUSAL, bool MXO = false>
__global__ __launch_bounds__(256) void attn2_kernel(const lp_t* __restrict__ qkv, lp_t* __restrict__ out, int S, int H,
                                                    float scale_log2e, int grp_R0, int grp_Lc, int nqb, uint8_t* __restrict__ mxs = nullptr,
                                                    int m128 = 0) {
  // Grouped sequences (causal only; grp_R0 > 0): rows [0, grp_Lc) are a SHARED prefix, rows [grp_R0, S) are independent 32-row
  // suffix blocks (one per search target) that each attend to the shared prefix and, causally, to themselves — the prefill of
  // T prompts that share their first grp_Lc tokens, with the prefix's K/V computed once (engine.hip::score_grouped).  A query
  // block inside the suffix region walks the prefix's key tiles (keys >= grp_Lc masked) and then its own 128 rows, where wave w
  // only looks at its own 32-row sub-tile.
  constexpr int KS = D / 16, DB = D / 32;
  constexpr int KT = 64;                            // keys per tile (32 with a 4-deep ring of the same bytes measured no better)
  constexpr int NBUF = 128 / KT;                    // ring depth (the ring always holds 128 keys of K and of V)
  constexpr int KBYTES = KT * D * 2;                // K tile = V tile bytes
  constexpr int KCH = D / 8;                        // 16-B chunks per K row
  constexpr int KROWS_PER_INST = 64 / KCH;          // K rows covered by one wave-wide DMA instruction
  constexpr int K_INST = KT / KROWS_PER_INST / 4;   // K DMA instructions per wave per tile (4 waves)
  static_assert(K_INST >= 1, "tile too small for 4 DMA waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // XCD-aware block order (1-D grid; the hardware deals workgroup L to XCD L % 8): each XCD gets a CONTIGUOUS range of
  // (sequence, head, query-block) triples with the query block fastest, so the query blocks of one head — which all walk the
  // same K/V rows — run next to each other on ONE XCD and share its L2 instead of fetching K/V into up to eight L2s.
  int h, b, qblk;
  {
    const int total = (int)gridDim.x, L = (int)blockIdx.x;
    const int per = total >> 3, rem = total & 7, x = L & 7;
    const int v = (x < rem ? x * (per + 1) : rem * (per + 1) + (x - rem) * per) + (L >> 3);
    qblk = v % nqb;
    const int hb = v / nqb;
    h = hb % H;
    b = hb / H;
  }
  // causal: the last query block has the most key tiles — launch the heavy blocks first so the tail of the grid is light
  const int q0b = (CAUSAL ? nqb - 1 - qblk : qblk) * 128;
  const int q0 = q0b + wave * 32;
  const bool active = q0 < S;
  const int qi = lane & 31, h2 = lane >> 5;
  const int query = q0 + qi;
  const int qrow = query < S ? query : S - 1;
  const int64_t ld = 3 * (int64_t)H * D;
  const lp_t* Kg = qkv + (int64_t)b * S * ld + (int64_t)H * D + h * D;
  const lp_t* Vg = qkv + (int64_t)b * S * ld + 2 * (int64_t)H * D + h * D;      // V rows of the fused qkv buffer

  const bool sfx = CAUSAL && grp_R0 > 0 && q0b >= grp_R0;          // block-uniform: this query block lies in the suffix region
  const int nsh = sfx ? (grp_Lc + KT - 1) / KT : 0;                   // key tiles of the shared prefix
  auto key0 = [&](int t) { return sfx ? (t < nsh ? t * KT : q0b + (t - nsh) * KT) : t * KT; };

  // per-lane DMA sources.  Round 6: buffer loads (SRD over the K / V columns of this sequence, per-lane byte offset fixed for the
  // whole block, the tile's first key as the SCALAR offset) instead of one 64-bit address computation per instruction — the eight DMA
  // instructions of a tile cost 1365 cycles of issue in round 5's timeline (profiles/r05_attn_timeline.txt), mostly VALU address
  // arithmetic — and rows past the sequence's end fall outside the descriptor's range and read as ZERO (their probabilities are
  // exactly 0 either way: masked keys) instead of being clamped per lane.
#ifndef ATTN_BUFFER_DMA
#define ATTN_BUFFER_DMA 1
#endif
  const int64_t ldb = ld * (int64_t)sizeof(lp_t);
  uint32_t koffs[K_INST], voffs[K_INST];
  const lp_t* ksrc[K_INST];
  int krow_l[K_INST];
#pragma unroll
  for (int i = 0; i < K_INST; ++i) {
    const int row = (i * 4 + wave) * KROWS_PER_INST + lane / KCH;       // key row inside the tile
    const int ch = lane % KCH;
    const int sw = (D == 64) ? ((row >> 1) & 7) : (row & 15);
    krow_l[i] = row;
    ksrc[i] = Kg + (ch ^ sw) * 8;
    koffs[i] = (uint32_t)(row * ldb + (ch ^ sw) * 16);
  }
  // V: the tile as it lies in the qkv buffer, [64 keys][D], DMA'd like the K tile and transposed on the way OUT of LDS by
  // ds_read_b64_tr_b16.  Its 16-byte chunks are swizzled per QUAD of chunks (64 bytes = what one half-wave reads of a key
  // row): quad' = quad ^ g(key), g = (key >> 1) & 1 for 128-byte rows and key & 3 for 256-byte rows, so that the 4 key rows
  // one transpose read touches fall into different bank groups.
  const lp_t* vsrc[K_INST];
#pragma unroll
  for (int i = 0; i < K_INST; ++i) {
    const int row = (i * 4 + wave) * KROWS_PER_INST + lane / KCH;
    const int ch = lane % KCH;
    const int gq = (D == 64) ? ((row >> 1) & 1) : (row & 3);
    vsrc[i] = Vg + (ch ^ (gq << 2)) * 8;
    voffs[i] = (uint32_t)(row * ldb + (ch ^ (gq << 2)) * 16);
  }
  // valid bytes from the first K (V) element of this sequence and head: rows 0 .. S - 1, D elements of the last row
  const uint32_t span = (uint32_t)((int64_t)(S - 1) * ldb + D * (int)sizeof(lp_t));
#if defined(__HIP_DEVICE_COMPILE__)      // (the resource type does not exist in the host pass, which would silently drop the kernel's stub)
  const __amdgpu_buffer_rsrc_t krsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Kg, 0, (int)span, 0x00020000);
  const __amdgpu_buffer_rsrc_t vrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Vg, 0, (int)span, 0x00020000);
#endif
  auto stage = [&](int t) {
    char* base = smem + (t % NBUF) * 2 * KBYTES;
    const int kt0 = key0(t);
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (ATTN_BUFFER_DMA) {
      const int soff = __builtin_amdgcn_readfirstlane((int)(kt0 * ldb));
#pragma unroll
      for (int i = 0; i < K_INST; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(krsrc, (lptr_t)(base + (i * 4 + wave) * 1024), 16, koffs[i], soff, 0, 0);
#pragma unroll
      for (int i = 0; i < K_INST; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(vrsrc, (lptr_t)(base + KBYTES + (i * 4 + wave) * 1024), 16, voffs[i], soff, 0, 0);
      return;
    }
#endif
#pragma unroll
    for (int i = 0; i < K_INST; ++i) {
      int kr = kt0 + krow_l[i];
      kr = kr < S ? kr : S - 1;
      __builtin_amdgcn_global_load_lds((gptr_t)(ksrc[i] + (int64_t)kr * ld), (lptr_t)(base + (i * 4 + wave) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < K_INST; ++i) {
      int kr = kt0 + krow_l[i];
      kr = kr < S ? kr : S - 1;                      // rows past S: their probabilities are exactly 0
      __builtin_amdgcn_global_load_lds((gptr_t)(vsrc[i] + (int64_t)kr * ld), (lptr_t)(base + KBYTES + (i * 4 + wave) * 1024), 16, 0,
                                       0);
    }
  };
  // transpose-read addressing (ds_read_b64_tr_b16: inside a 16-lane group lane t supplies the 8-byte piece
  // V[key0 + t/4][d0 + 4*(t%4) ..+3] and receives V[key0 + 0..3][d0 + t]): lane = (MFMA row d = lane % 32, key half h2)
  typedef __attribute__((ext_vector_type(4))) short s4_t;
  typedef __attribute__((address_space(3))) s4_t* lds_s4_t;
  const int t16 = lane & 15, g16 = (lane >> 4) & 1;
  const int tr_gq = (D == 64) ? ((t16 >> 3) & 1) : (t16 >> 2);      // g(key) of this lane's key row (key0 % 4 == 0)
  const int tr_off = ((t16 >> 2) + 4 * h2) * (D * 2) + (2 * g16 + ((t16 & 3) >> 1)) * 16 + (t16 & 1) * 8;

  // Q fragments, PRE-SCALED by scale*log2(e) (rounded to the storage type once): the QK^T accumulator then already holds the
  // scores in exp2 units, and with the running reference max handed to the MFMA as its C operand (negm, 16 registers all equal
  // to -m) the accumulator comes out as s - m — no per-score multiply/subtract on the VALU.
  const lp_t* Qp = qkv + ((int64_t)b * S + qrow) * ld + h * D + h2 * 8;
  lpx8 qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const lpx8 raw = *(const lpx8*)(Qp + ks * 16);
#pragma unroll
    for (int e = 0; e < 8; ++e) qf[ks][e] = (short)f2lp(lp2f((lp_t)raw[e]) * scale_log2e);
  }
  f32x16 oacc[DB];
#pragma unroll
  for (int db = 0; db < DB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[db][r] = 0.f;
  f32x16 negm;
#pragma unroll
  for (int r = 0; r < 16; ++r) negm[r] = 0.f;
  float m = 0.f, l = 0.f;          // l: this lane's partial row sum (its 16 of every 32 keys); the lane^32 halves are added at the end
  bool first = true;               // wave-uniform: the first sub-tile a wave computes fixes m to a TRUE row maximum

  const int kend = CAUSAL ? min(S, q0b + 128) : S;
  const int nkt = sfx ? nsh + (kend - q0b + KT - 1) / KT : (kend + KT - 1) / KT;
  const int kswz = (D == 64) ? ((qi >> 1) & 7) : (qi & 15);   // key row = st*32 + qi: the st*32 term leaves both swizzles unchanged
  // per-lane LDS byte offsets inside a ring slot (everything else of an address is a compile-time immediate: the key loop is
  // unrolled over the ring slots and the two 32-key sub-tiles of a tile)
  int koff[KS], voff[DB];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) koff[ks] = qi * (D * 2) + (((ks * 2 + h2) ^ kswz) * 16);
#pragma unroll
  for (int db = 0; db < DB; ++db) voff[db] = (int)(uintptr_t)(lptr_t)smem + KBYTES + tr_off + ((db ^ tr_gq) * 64);

  auto subtile = [&](auto SLOT_, auto ST_, int t) {
    constexpr int SLOT = decltype(SLOT_)::value, st = decltype(ST_)::value;
    constexpr int SBASE = SLOT * 2 * KBYTES + st * 32 * (D * 2);
    const int kt0 = key0(t) + st * 32;
    const bool shared_tile = sfx && t < nsh;                // prefix keys seen from a suffix block: no causal mask, limit grp_Lc
    const int klimit = shared_tile ? grp_Lc : S;
    if (kt0 >= (shared_tile ? grp_Lc : kend)) return;
    if (sfx && !shared_tile && kt0 != q0) return;           // suffix block: of its own 128 rows a wave sees only its 32
    if (CAUSAL && !shared_tile && kt0 > q0 + 31) return;    // wave-uniform: sub-tile entirely above the diagonal
    // ---- S^T - m = K . Q'^T + (-m) ----
    f32x16 sacc = negm;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const lpx8 kf = *(const lpx8*)(smem + koff[ks] + SBASE);
      sacc = mfma_32x32x16(kf, qf[ks], sacc);
    }
    // V operand fragments of this sub-tile: keys 16j + 4*h2 + {0..3} (lo) and 16j + 8 + 4*h2 + {0..3} (hi) — the keys whose
    // probabilities this lane will hold in pb — by ds_read_b64_tr_b16 as INLINE ASM: the compiler guards its own builtin for
    // that instruction with `s_waitcnt vmcnt(0)` (it cannot tell the ring slot being read from the slot the in-flight LDS-DMA
    // of the NEXT tile writes), which made every tile wait for its successor's prefetch.  EARLY_V issues the reads before the
    // softmax so that they fly under it (8 * DB more live registers).
    s4_t vlo[2][DB], vhi[2][DB];
#define ATTN_V_READS()                                                                                                         \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) _Pragma("unroll") for (int db = 0; db < DB; ++db) {                              \
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vlo[j][db]) : "v"(voff[db]), "i"(SBASE + 16 * j * (D * 2)));       \
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vhi[j][db]) : "v"(voff[db]), "i"(SBASE + (16 * j + 8) * (D * 2))); \
  }
    constexpr bool EARLY_V = ATTN_EARLY_V(D);
    if constexpr (EARLY_V) { ATTN_V_READS() }
    // this lane: query column `query`, keys kt0 + (r&3) + 8*(r>>2) + 4*h2
    const bool causal_here = CAUSAL && !shared_tile;
    const bool need_mask = (kt0 + 32 > klimit) || (causal_here && kt0 + 31 > q0);
    if (need_mask) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kt0 + (r & 3) + 8 * (r >> 2) + 4 * h2;
        const bool masked = (key >= klimit) || (causal_here && key > query);
        sacc[r] = masked ? -1e30f : sacc[r];
      }
    }
    // ---- online softmax against the STALE reference m: p = exp2(s - m) straight from the accumulator.  m is only touched on
    // the slow path: the wave's first sub-tile (m := true row max, so the final row sum is >= 1) and whenever some row's
    // probabilities outgrow 2^16 (m := new row max, history rescaled).  exp2 stays finite below that, P is bf16 (relative
    // precision) and l / O are fp32, so nothing else depends on how stale m is.
    constexpr float BIG = 65536.0f;
    float p[16];
    float rs = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      p[r] = __builtin_amdgcn_exp2f(sacc[r]);
      rs += p[r];
    }
    if (first || __any(!(rs <= BIG))) {                               // wave-uniform
      float mx = sacc[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sacc[r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));                         // row max relative to the current m
      const float delta = first ? fmaxf(mx, -1e4f) : fmaxf(mx, 0.f);
      const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-delta);   // first: l = O = 0 (and exp2(-delta) may overflow: 0 * inf)
      l *= alpha;
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
      m += delta;
#pragma unroll
      for (int r = 0; r < 16; ++r) negm[r] = -m;
      rs = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        p[r] = __builtin_amdgcn_exp2f(sacc[r] - delta);               // masked: exp2(-1e30 - delta) = 0
        rs += p[r];
      }
      first = false;
    }
    l += rs;
    if constexpr (!EARLY_V) { ATTN_V_READS() }
#undef ATTN_V_READS
    // the transpose reads are invisible to the compiler's lgkmcnt bookkeeping: wait for them here (in-order counter)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int db = 0; db < DB; ++db) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vlo[j][db]), "+v"(vhi[j][db]));
    // ---- O^T += V^T . P^T ----
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      lpx8 pb;
#pragma unroll
      for (int i = 0; i < 8; ++i) pb[i] = (short)f2lp(p[8 * j + i]);
#pragma unroll
      for (int db = 0; db < DB; ++db) {
        const lpx8 vf = (lpx8){vlo[j][db][0], vlo[j][db][1], vlo[j][db][2], vlo[j][db][3],
                               vhi[j][db][0], vhi[j][db][1], vhi[j][db][2], vhi[j][db][3]};
        oacc[db] = mfma_32x32x16(vf, pb, oacc[db]);
      }
    }
  };

  // ONE barrier per tile: after it every wave has (a) seen its share of tile t land (vmcnt(0): tile t's DMA is the youngest
  // request) and (b) finished computing tile t-1, whose ring slot the tile requested next (t + 1) reuses.
  // (Two barriers per tile parked the waves 43 % of the time, rocprofv3 SQ_WAIT_ANY.)
  auto tile = [&](auto SLOT_, int t) {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (t + 1 < nkt) stage(t + 1);
    if (active) {
      subtile(SLOT_, std::integral_constant<int, 0>{}, t);
      subtile(SLOT_, std::integral_constant<int, 1>{}, t);
    }
  };
  stage(0);
  for (int t = 0; t < nkt; t += 2) {
    tile(std::integral_constant<int, 0>{}, t);
    if (t + 1 < nkt) tile(std::integral_constant<int, 1>{}, t + 1);
  }
  l += __shfl_xor(l, 32, 64);

  if constexpr (MXO) {
    // lane (query, h2) holds d = db * 32 + g * 8 + 4 * h2 + e of its row: a block of 32 d = this lane's 16 values + lane ^ 32's
    const float inv = 1.0f / l;
    const int row = b * S + query;
    uint8_t* op8 = (uint8_t*)out + (int64_t)row * ((int64_t)H * D) + h * D + 16 * h2;
#pragma unroll
    for (int db = 0; db < DB; ++db) {
      float f[16], mx = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        f[r] = rlp(oacc[db][r] * inv);
        mx = fmaxf(mx, fabsf(f[r]));
      }
      mx = mx_max_halves(mx);                       // (every lane takes part)
      const uint32_t e8 = mx_e8m0(mx);
      const float is = mx_inv_scale(e8);
      // this lane's four 4-byte pieces sit at d = g * 8 + 4 * h2; two half swaps with lane ^ 32 turn them into 16 CONSECUTIVE bytes per
      // lane (h2 = 0: bytes 0..15 of the block, h2 = 1: 16..31) — one 16-byte store instead of four 4-byte ones
      uint32_t pk[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) pk[g] = mx_pack4(f[g * 4] * is, f[g * 4 + 1] * is, f[g * 4 + 2] * is, f[g * 4 + 3] * is);
      const auto s02 = __builtin_amdgcn_permlane32_swap(pk[0], pk[2], false, false);
      const auto s13 = __builtin_amdgcn_permlane32_swap(pk[1], pk[3], false, false);
      if (active && query < S) {
        typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
        const u32x4 o = {s02[0], s02[1], s13[0], s13[1]};
        *(u32x4*)(op8 + db * 32) = o;
        if (h2 == 0) mxs[mx_scale_offset(row, h * DB + db, m128)] = (uint8_t)e8;
      }
    }
    return;
  }
#ifndef ATTN_WIDE_STORES
#define ATTN_WIDE_STORES 1
#endif
  if constexpr (ATTN_WIDE_STORES) {
    // Round 6: this lane's 8-byte pieces (d = db * 32 + g * 8 + 4 * h2 + e) and lane ^ 32's interleave; two half swaps per dword turn
    // them into 16 consecutive d per lane (h2 = 0: d = db * 32 + 0..15, h2 = 1: 16..31) = two 16-byte stores instead of four 8-byte
    // ones — the store tail of a block is issue-bound (MI355X guide T21).  Same values, same addresses.
    const float inv = 1.0f / l;
    lp_t* op = out + ((int64_t)b * S + query) * ((int64_t)H * D) + h * D + 16 * h2;
    const bool wr = active && query < S;
#pragma unroll
    for (int db = 0; db < DB; ++db) {
      uint32_t pk[4][2];
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int k = 0; k < 2; ++k)
          pk[g][k] = (uint32_t)(uint16_t)f2lp(oacc[db][g * 4 + 2 * k] * inv) | ((uint32_t)(uint16_t)f2lp(oacc[db][g * 4 + 2 * k + 1] * inv) << 16);
      typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {       // (g = gp, g = gp + 2): h2 = 0 keeps group gp of both lanes, h2 = 1 group gp + 2
        const auto s0 = __builtin_amdgcn_permlane32_swap(pk[gp][0], pk[gp + 2][0], false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(pk[gp][1], pk[gp + 2][1], false, false);
        const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
        if (wr) *(u32x4*)(op + db * 32 + gp * 8) = o;
      }
    }
    return;
  }
  if (active && query < S) {
    const float inv = 1.0f / l;
    lp_t* op = out + ((int64_t)b * S + query) * ((int64_t)H * D) + h * D + 4 * h2;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        lpx4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (short)f2lp(oacc[db][g * 4 + e] * inv);
        *(lpx4*)(op + db * 32 + g * 8) = o;
      }
  }
}

// ---------------- small generic attention (SAM head): one wave per (b, head, query) ----------------
template <int D>
__global__ __launch_bounds__(256) void small_attn_kernel(const lp_t* __restrict__ q, const lp_t* __restrict__ k,
                                                         const lp_t* __restrict__ v, lp_t* __restrict__ out, int B,
                                                         int Nq, int Nk, int H, float scale) {
  const int lane = threadIdx.x & 63;
  const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= (int64_t)B * H * Nq) return;
  const int qi = (int)(wid % Nq);
  const int h = (int)((wid / Nq) % H);
  const int b = (int)(wid / ((int64_t)Nq * H));
  const int C = H * D;
  float qv[D];
  const lp_t* qp = q + ((int64_t)b * Nq + qi) * C + h * D;
#pragma unroll
  for (int d = 0; d < D; ++d) qv[d] = lp2f(qp[d]);
  // pass 1: scores for this lane's keys (kept for up to 40 keys per lane = 2560 keys)
  constexpr int MAXK = 40;
  float sc[MAXK];
  float mx = -1e30f;
#pragma unroll
  for (int i = 0; i < MAXK; ++i) {
    const int key = i * 64 + lane;
    float s = -1e30f;
    if (key < Nk) {
      const lp_t* kp = k + ((int64_t)b * Nk + key) * C + h * D;
      float a = 0.f;
#pragma unroll
      for (int d = 0; d < D; ++d) a += qv[d] * lp2f(kp[d]);
      s = rlp(rlp(a) * scale);     // reference: bf16 matmul, then bf16 divide by sqrt(d)
    }
    sc[i] = s;
    mx = fmaxf(mx, s);
  }
  mx = wave_max(mx);
  float o[D];
#pragma unroll
  for (int d = 0; d < D; ++d) o[d] = 0.f;
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < MAXK; ++i) {
    const int key = i * 64 + lane;
    if (key < Nk) {
      const float e = __expf(sc[i] - mx);
      sum += e;
      sc[i] = e;
    }
  }
  sum = wave_sum(sum);
  const float inv = 1.0f / sum;
#pragma unroll
  for (int i = 0; i < MAXK; ++i) {
    const int key = i * 64 + lane;
    if (key < Nk) {
      const float pr = rlp(sc[i] * inv);   // softmax output is bf16 in the reference
      const lp_t* vp = v + ((int64_t)b * Nk + key) * C + h * D;
#pragma unroll
      for (int d = 0; d < D; ++d) o[d] += pr * lp2f(vp[d]);
    }
  }
#pragma unroll
  for (int d = 0; d < D; ++d) o[d] = wave_sum(o[d]);
  if (lane == 0) {
    lp_t* op = out + ((int64_t)b * Nq + qi) * C + h * D;
#pragma unroll
    for (int d = 0; d < D; ++d) op[d] = f2lp(o[d]);
  }
}

// few-keys variant (image -> token cross attention of the SAM head: 2304 queries x 6 keys): one THREAD per (b, head, query)
template <int D, int MAXK>
__global__ __launch_bounds__(256) void small_attn_fewkeys_kernel(const lp_t* __restrict__ q, const lp_t* __restrict__ k,
                                                                 const lp_t* __restrict__ v, lp_t* __restrict__ out,
                                                                 int B, int Nq, int Nk, int H, float scale) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)B * H * Nq) return;
  const int h = (int)(idx % H);                      // heads fastest: neighbouring threads read neighbouring 32-byte slices
  const int qi = (int)((idx / H) % Nq);
  const int b = (int)(idx / ((int64_t)H * Nq));
  const int C = H * D;
  const lp_t* qp = q + ((int64_t)b * Nq + qi) * C + h * D;
  float qv[D];
#pragma unroll
  for (int d0 = 0; d0 < D; d0 += 8) {
    const lpx8 t = *(const lpx8*)(qp + d0);
#pragma unroll
    for (int e = 0; e < 8; ++e) qv[d0 + e] = lp2f((lp_t)t[e]);
  }
  float sc[MAXK];
  float mx = -1e30f;
#pragma unroll
  for (int j = 0; j < MAXK; ++j) {
    sc[j] = -1e30f;
    if (j < Nk) {
      const lp_t* kp = k + ((int64_t)b * Nk + j) * C + h * D;
      float a = 0.f;
#pragma unroll
      for (int d0 = 0; d0 < D; d0 += 8) {
        const lpx8 t = *(const lpx8*)(kp + d0);
#pragma unroll
        for (int e = 0; e < 8; ++e) a += qv[d0 + e] * lp2f((lp_t)t[e]);
      }
      sc[j] = rlp(rlp(a) * scale);
      mx = fmaxf(mx, sc[j]);
    }
  }
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < MAXK; ++j) {
    sc[j] = j < Nk ? __expf(sc[j] - mx) : 0.f;
    sum += sc[j];
  }
  const float inv = 1.0f / sum;
  float o[D];
#pragma unroll
  for (int d = 0; d < D; ++d) o[d] = 0.f;
#pragma unroll
  for (int j = 0; j < MAXK; ++j) {
    if (j < Nk) {
      const float pr = rlp(sc[j] * inv);
      const lp_t* vp = v + ((int64_t)b * Nk + j) * C + h * D;
#pragma unroll
      for (int d0 = 0; d0 < D; d0 += 8) {
        const lpx8 t = *(const lpx8*)(vp + d0);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[d0 + e] += pr * lp2f((lp_t)t[e]);
      }
    }
  }
  lp_t* op = out + ((int64_t)b * Nq + qi) * C + h * D;
#pragma unroll
  for (int d0 = 0; d0 < D; d0 += 8) {
    lpx8 t;
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = (short)f2lp(o[d0 + e]);
    *(lpx8*)(op + d0) = t;
  }
}

}  // namespace

hipError_t attn_prepare(lp_t* qkv, const lp_t* cos_sin, int B, int S, int H, int D, hipStream_t s, int grp_R0, int grp_Lc) {
  if (D != 64 && D != 128) return hipErrorInvalidValue;
  if (cos_sin) {
    const int64_t n = (int64_t)B * S * 2 * H * (D / 16);
    hipLaunchKernelGGL(rope_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, qkv, cos_sin, B * S, S, H, D, grp_R0, grp_Lc);
  }
  return hipGetLastError();
}

template <int D, bool CAUSAL, bool MXO = false>
static hipError_t launch_attn2(const lp_t* qkv, lp_t* out, int B, int S, int H, float sl, hipStream_t s, int grp_R0 = 0,
                               int grp_Lc = 0, uint8_t* mxs = nullptr) {
  constexpr int LDS = 4 * 64 * D * 2;
  static bool attr_done = false;
  auto kern = attn2_kernel<D, CAUSAL, MXO>;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  const int nqb = (S + 127) / 128;
  dim3 grid((unsigned)(nqb * H * B));
  hipLaunchKernelGGL(kern, grid, dim3(256), LDS, s, qkv, out, S, H, sl, grp_R0, grp_Lc, nqb, mxs, (B * S) >> 7);
  return hipGetLastError();
}

hipError_t attn_forward(const lp_t* qkv, lp_t* out, int B, int S, int H, int D, int causal, float scale, hipStream_t s, int grp_R0,
                        int grp_Lc) {
  if (D != 64 && D != 128) return hipErrorInvalidValue;
  if (grp_R0 > 0 && (!causal || D != 128 || grp_R0 % 128 || grp_Lc <= 0 || grp_Lc > grp_R0 || (S - grp_R0) % 32)) return hipErrorInvalidValue;
  const float sl = scale * 1.4426950408889634f;
  if (D == 64) return causal ? launch_attn2<64, true>(qkv, out, B, S, H, sl, s) : launch_attn2<64, false>(qkv, out, B, S, H, sl, s);
  return causal ? launch_attn2<128, true>(qkv, out, B, S, H, sl, s, grp_R0, grp_Lc) : launch_attn2<128, false>(qkv, out, B, S, H, sl, s);
}

hipError_t attn_forward_mx(const lp_t* qkv, uint8_t* out8, uint8_t* scales, int B, int S, int H, float scale, hipStream_t s, int grp_R0,
                           int grp_Lc) {
#ifdef VSTAR_LP_F16
  return hipErrorInvalidValue;
#else
  if ((int64_t)B * S % 128 || !out8 || !scales) return hipErrorInvalidValue;
  if (grp_R0 > 0 && (grp_R0 % 128 || grp_Lc <= 0 || grp_Lc > grp_R0 || (S - grp_R0) % 32)) return hipErrorInvalidValue;
  return launch_attn2<128, true, true>(qkv, (lp_t*)out8, B, S, H, scale * 1.4426950408889634f, s, grp_R0, grp_Lc, scales);
#endif
}

hipError_t small_attention(const lp_t* q, const lp_t* k, const lp_t* v, lp_t* out, int B, int Nq, int Nk, int H,
                           int D, hipStream_t s) {
  if (Nk > 40 * 64) return hipErrorInvalidValue;
  const int64_t waves = (int64_t)B * H * Nq;
  dim3 grid((unsigned)((waves + 3) / 4));
  const float scale = 1.0f / sqrtf((float)D);
  if (Nk <= 8 && (D == 16 || D == 32)) {
    const int64_t threads = (int64_t)B * H * Nq;
    dim3 g((unsigned)((threads + 255) / 256));
    if (D == 16) hipLaunchKernelGGL((small_attn_fewkeys_kernel<16, 8>), g, dim3(256), 0, s, q, k, v, out, B, Nq, Nk, H, scale);
    else hipLaunchKernelGGL((small_attn_fewkeys_kernel<32, 8>), g, dim3(256), 0, s, q, k, v, out, B, Nq, Nk, H, scale);
    return hipGetLastError();
  }
  if (D == 32) hipLaunchKernelGGL(small_attn_kernel<32>, grid, dim3(256), 0, s, q, k, v, out, B, Nq, Nk, H, scale);
  else if (D == 16) hipLaunchKernelGGL(small_attn_kernel<16>, grid, dim3(256), 0, s, q, k, v, out, B, Nq, Nk, H, scale);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

}  // namespace VS_NS
