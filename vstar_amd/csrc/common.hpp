// common.hpp — shared device/host helpers for the gfx950 kernels (wave64, bf16 storage, fp32 math).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define WAVE 64

__device__ __host__ __forceinline__ float bf2f(bf16_t v) {
  union { uint32_t u; float f; } x;
  x.u = ((uint32_t)v) << 16;
  return x.f;
}
// round-to-nearest-even, NaN preserved (matches torch's float -> bfloat16).  Device code uses the gfx950 hardware
// conversion (v_cvt_pk_bf16_f32); the bit-twiddling form is the host path (weight packing).
__device__ __host__ __forceinline__ bf16_t f2bf(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
  const __bf16 b = (__bf16)f;
  return __builtin_bit_cast(unsigned short, b);
#endif
  union { uint32_t u; float f; } x;
  x.f = f;
  if ((x.u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((x.u >> 16) | 0x40);
  uint32_t lsb = (x.u >> 16) & 1u;
  x.u += 0x7fffu + lsb;
  return (bf16_t)(x.u >> 16);
}
__device__ __forceinline__ float rbf(float f) { return bf2f(f2bf(f)); }  // round through bf16

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

// activations, with the reference's bf16 rounding points (Appendix C of SURVEY.md)
__device__ __forceinline__ float act_quick_gelu_bf16(float t) {  // t already bf16-rounded
  float u = rbf(1.702f * t);
  float s = rbf(1.0f / (1.0f + __expf(-u)));
  return t * s;
}
__device__ __forceinline__ float act_gelu_erf(float t) { return 0.5f * t * (1.0f + erff(t * 0.70710678118654752f)); }
__device__ __forceinline__ float act_silu_bf16(float g) {  // g already bf16-rounded; torch silu on bf16 rounds once
  return rbf(g / (1.0f + __expf(-g)));
}
