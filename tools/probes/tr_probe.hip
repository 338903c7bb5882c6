// Probe of ds_read_b64_tr_b16 (gfx950): every lane supplies the address of its own 8-byte piece; LDS holds value = index.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probes/tr_probe.hip -o tr_probe ; prints what each lane received.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) short s4;
__global__ void k(short* out) {
  __shared__ short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
  __syncthreads();
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(lds + threadIdx.x * 4));
  *(s4*)(out + threadIdx.x * 4) = v;
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int ok = 1;
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int j = 0; j < 4; ++j) {
      printf(" %4d", h[l * 4 + j]);
      const int g = l / 16, i = l % 16;
      if (h[l * 4 + j] != (g * 16 + 4 * j + i / 4) * 4 + i % 4) ok = 0;
    }
    printf("\n");
  }
  printf("hypothesis result(lane i, elem j) = piece(lane 4j + i/4)[i%%4]: %s\n", ok ? "CONFIRMED" : "REFUTED");
  return 0;
}
