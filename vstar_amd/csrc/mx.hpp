// mx.hpp — block-scaled (OCP MX style) fp8 activations for the W8A8 mode (BASELINE config 5, round 6).
//
// An activation row is cut into blocks of 32 consecutive k; block values are stored as OCP e4m3 bytes of x / 2^(e - 127) with ONE E8M0
// byte e per block, chosen as the SMALLEST power of two that brings the block's largest magnitude to <= 448 (no saturation; the OCP
// recipe floor(log2 amax) - 8 would clip (448, 512) x 2^n).  v_mfma_scale_f32_16x16x128_f8f6f4 applies 2^(e - 127) per (row, 32 k) inside
// the matrix pipe (profiles/r06_mfma_scale_probe.txt), so the consumer GEMM needs no per-token scale and the PRODUCER can quantise a
// block the moment it holds its 32 values — in the gate|up epilogue (SiLU * up) and in the attention epilogue — which removes the two
// stand-alone per-token quantisation passes of the round-3 scheme.  The reference has no fp8 path: oracle/vsm_oracle.py::mx_fake_quant
// restates this file's arithmetic.
//
// Scale bytes are laid out for the consumer (gemm4w.hip, MX text): for K-tile T (128 k) and row block R (128 rows) 512 bytes
// [lane = (k-block in tile) * 16 + row % 16][fragment m = (row % 128) / 16], K-tile major, so that the 256 rows of a GEMM tile are ONE
// KiB per K-tile (a 17th DMA piece) and every lane finds its eight bytes — one per A fragment — with one ds_read_b64.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

namespace VS_NS {

// byte offset of the scale of (row, k-block kb = k / 32); m128 = rows / 128 of the activation matrix (rows % 128 == 0)
__host__ __device__ inline int64_t mx_scale_offset(int row, int kb, int m128) {
  return ((int64_t)(kb >> 2) * m128 + (row >> 7)) * 512 + ((kb & 3) * 16 + (row & 15)) * 8 + ((row >> 4) & 7);
}
inline size_t mx_scale_bytes(int64_t rows, int64_t cols) { return (size_t)(rows * (cols / 32)); }

// E8M0 byte for a block whose largest magnitude is amax (>= 0, finite): with amax = 1.f x 2^E, e = E + 127 - 8, one more when
// 1.f > 1.75 (448 = 1.75 x 2^8); integer arithmetic on the bits, so the oracle restates it exactly
__device__ __forceinline__ uint32_t mx_e8m0(float amax) {
  const uint32_t b = __float_as_uint(amax);
  const int e = (int)(b >> 23) - 8 + ((b & 0x7fffffu) > 0x600000u ? 1 : 0);
  return (uint32_t)(e < 0 ? 0 : e);
}
// 2^(127 - e): multiplying by it is exact
__device__ __forceinline__ float mx_inv_scale(uint32_t e) { return __uint_as_float((254u - e) << 23); }

// all-reduce max of a NON-NEGATIVE float (compared as its bits) over lane ^ 32 / over lane ^ 16 (neighbouring 16-lane rows of a
// wave): gfx950's v_permlane32_swap / v_permlane16_swap — plain VALU instructions instead of two trips through the LDS crossbar
__device__ __forceinline__ float mx_max_halves(float v) {
  const uint32_t u = __float_as_uint(v);
  const auto a = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(a[0] > a[1] ? a[0] : a[1]);
}
__device__ __forceinline__ float mx_max_row_pairs(float v) {      // lane ^ 16
  const uint32_t u = __float_as_uint(v);
  const auto b = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return __uint_as_float(b[0] > b[1] ? b[0] : b[1]);
}

__device__ __forceinline__ uint32_t mx_pack4(float a, float b, float c, float d) {
  uint32_t v = 0;
  v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, v, false);
  v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
  return v;
}

}  // namespace VS_NS
