// common.hpp — shared device/host helpers for the gfx950 kernels (wave64, 16-bit storage, fp32 math).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Every kernel file is dtype-generic over its 16-bit storage element `lp_t`: the translation unit is compiled once for
// bf16 (default; the VSM path, torch_dtype=bfloat16 at visual_search.py:145) and once with -DVSTAR_LP_F16 for IEEE fp16 (the
// VQA-LLM path, torch_dtype=float16 at LLaVA/llava/model/builder.py:43).  The storage type selects the conversion
// instructions, the MFMA opcode and the namespace the host launchers live in; everything else (DMA, LDS layouts,
// schedules) is identical because both are 2-byte elements.
#ifdef VSTAR_LP_F16
#define VS_NS vs_f16
#define VS_DTYPE_NAME "f16"
#else
#define VS_NS vs_bf16
#define VS_DTYPE_NAME "bf16"
#endif

typedef uint16_t lp_t;  // raw bits of one low-precision element
typedef __attribute__((ext_vector_type(8))) short lpx8;
typedef __attribute__((ext_vector_type(4))) short lpx4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define WAVE 64

namespace VS_NS {

#ifdef VSTAR_LP_F16
__device__ __host__ __forceinline__ float lp2f(lp_t v) { return (float)__builtin_bit_cast(_Float16, v); }
// round-to-nearest-even like torch's float -> half (v_cvt_f16_f32 on the device)
__device__ __host__ __forceinline__ lp_t f2lp(float f) { return __builtin_bit_cast(unsigned short, (_Float16)f); }
typedef __attribute__((ext_vector_type(8))) _Float16 mfma_x8;
__device__ __forceinline__ f32x4 mfma_16x16x32(lpx8 a, lpx8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(mfma_x8, a), __builtin_bit_cast(mfma_x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma_32x32x16(lpx8 a, lpx8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(mfma_x8, a), __builtin_bit_cast(mfma_x8, b), c, 0, 0, 0);
}
#else
__device__ __host__ __forceinline__ float lp2f(lp_t v) {
  union { uint32_t u; float f; } x;
  x.u = ((uint32_t)v) << 16;
  return x.f;
}
// round-to-nearest-even, NaN preserved (matches torch's float -> bfloat16).  Device code uses the gfx950 hardware
// conversion (v_cvt_pk_bf16_f32); the bit-twiddling form is the host path (weight packing).
__device__ __host__ __forceinline__ lp_t f2lp(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
  const __bf16 b = (__bf16)f;
  return __builtin_bit_cast(unsigned short, b);
#endif
  union { uint32_t u; float f; } x;
  x.f = f;
  if ((x.u & 0x7fffffffu) > 0x7f800000u) return (lp_t)((x.u >> 16) | 0x40);
  uint32_t lsb = (x.u >> 16) & 1u;
  x.u += 0x7fffu + lsb;
  return (lp_t)(x.u >> 16);
}
__device__ __forceinline__ f32x4 mfma_16x16x32(lpx8 a, lpx8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma_32x32x16(lpx8 a, lpx8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
#endif
// v + (v of the lane at xor 1 / xor 2 inside a quad, or of the mirrored lane inside a group of 8) by DPP: no LDS traffic.
// After the xor-1 and xor-2 steps the four lanes of a quad agree, so the half-mirror step equals an xor-4 exchange.
__device__ __forceinline__ float dpp_add_xor1(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_add_xor2(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_add_half_mirror(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));
}
__device__ __forceinline__ float rlp(float f) { return lp2f(f2lp(f)); }  // round through the storage type

// fp8 e4m3 (OCP) x fp8 e4m3 -> fp32, K = 128, block scales fixed at 1.0 (E8M0 0x7F): operands are 32 bytes per lane, any
// k order as long as both operands use the same one (the dot product does not care)
typedef __attribute__((ext_vector_type(8))) int i32x8;
__device__ __forceinline__ f32x4 mfma_16x16x128_fp8(lpx8 a_lo, lpx8 a_hi, lpx8 b_lo, lpx8 b_hi, f32x4 c) {
  typedef __attribute__((ext_vector_type(4))) int i32x4;
  const i32x4 al = __builtin_bit_cast(i32x4, a_lo), ah = __builtin_bit_cast(i32x4, a_hi);
  const i32x4 bl = __builtin_bit_cast(i32x4, b_lo), bh = __builtin_bit_cast(i32x4, b_hi);
  const i32x8 a = {al[0], al[1], al[2], al[3], ah[0], ah[1], ah[2], ah[3]};
  const i32x8 b = {bl[0], bl[1], bl[2], bl[3], bh[0], bh[1], bh[2], bh[3]};
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// activations, with the reference's bf16 rounding points (Appendix C of SURVEY.md).  sigmoid = v_exp_f32 + v_add + v_rcp_f32
// (1 ulp, then rounded to 16 bits): an IEEE-exact fp32 divide costs ~10 VALU instructions per element, which made the SiLU /
// quick-GELU epilogues of the gate|up and ViT fc1 GEMMs VALU-bound (6 % and 30 % of those kernels)
#ifdef VSTAR_EXACT_SIGMOID   // bisect builds only (tools/noise_study.py): IEEE divide + expf instead of v_rcp_f32 / v_exp_f32
__device__ __forceinline__ float fast_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
#else
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
#endif
__device__ __forceinline__ float act_quick_gelu_bf16(float t) {  // t already bf16-rounded
  float u = rlp(1.702f * t);
  float s = rlp(fast_sigmoid(u));
  return t * s;
}
__device__ __forceinline__ float act_gelu_erf(float t) { return 0.5f * t * (1.0f + erff(t * 0.70710678118654752f)); }
__device__ __forceinline__ float act_silu_bf16(float g) {  // g already bf16-rounded; torch silu on bf16 rounds once
  return rlp(g * fast_sigmoid(g));
}

}  // namespace VS_NS
using namespace VS_NS;
