// dma_probe.hip — how fast can one CU pull operand tiles?  (round 3: is the GEMM K-loop bound by the LDS-DMA path?)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/dma_probe.hip -o /tmp/dma_probe && /tmp/dma_probe
// Every workgroup streams `iters` x 32 KiB tiles (256 rows x 128 B, row pitch `pitch`) either with global_load_lds (16 B/lane,
// LDS-DMA, as the GEMMs stage their K-tiles) or with plain global_load_dwordx4 into registers; the per-workgroup footprint
// (`span` rows) decides where the data comes from (64 KiB: L1/L2-resident; 4 MiB/workgroup: L2 / Infinity Cache; more: HBM).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((ext_vector_type(4))) int i32x4;

template <int MODE, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void probe(const char* src, long pitch, int rows_span, int iters, int* sink, long wg_stride) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const char* base = src + (long)blockIdx.x * wg_stride;
  // a "K-tile" = 256 rows x 128 B = 32 pieces of 8 rows x 128 B = 1 KiB; wave w moves pieces w, w+WAVES, ...
  const int st_r = lane >> 3, st_c = lane & 7;
  i32x4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    const long k0 = (long)(it % (int)(pitch / 128)) * 128;          // walk along the row like a K loop
    const int row_base = (it / (int)(pitch / 128)) * 256 % rows_span;
    char* lbase = smem + (it & 1) * 32768;
#pragma unroll
    for (int p = 0; p < 32 / WAVES; ++p) {
      const int piece = p * WAVES + wave;
      const int row = (row_base + piece * 8 + st_r) % rows_span;
      const char* g = base + (long)row * pitch + k0 + st_c * 16;
      if (MODE == 0) {
        __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(lbase + piece * 1024), 16, 0, 0);
      } else if (MODE == 2) {      // output side: 16-byte non-temporal row stores, as the GEMM epilogue writes C
        __builtin_nontemporal_store(acc, (i32x4*)g);
      } else if (MODE == 3) {
        *(i32x4*)g = acc;
      } else {
        const i32x4 v = *(const i32x4*)g;
        acc += v;
      }
    }
    if (MODE == 0) {
      // keep at most one tile's pieces in flight behind the current one (like a 2-deep ring)
      if (32 / WAVES == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (32 / WAVES == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (MODE != 0 && acc[0] + acc[1] + acc[2] + acc[3] == 0x12345678) sink[0] = 1;
  if (MODE == 0 && smem[threadIdx.x] == 77 && iters < 0) sink[1] = 1;
}

template <int MODE, int WAVES>
void run(const char* name, const char* buf, long pitch, int rows_span, int nwg, int* sink, long wg_stride, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto k = probe<MODE, WAVES>;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k, dim3(nwg), dim3(WAVES * 64), 65536, 0, buf, pitch, rows_span, iters, sink, wg_stride);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(nwg), dim3(WAVES * 64), 65536, 0, buf, pitch, rows_span, iters, sink, wg_stride);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)nwg * iters * 32768.0;
  printf("%-34s wgs %4d waves %d  pitch %6ld span %5d rows : %8.1f us  %7.2f TB/s  %6.1f KB/us/WG\n", name, nwg, WAVES, pitch, rows_span,
         ms * 1e3, bytes / ms / 1e9, 32.768 * iters / (ms * 1e3));
}

int main(int argc, char** argv) {
  const size_t total = (size_t)3 << 30;
  char* buf; int* sink;
  hipMalloc(&buf, total); hipMalloc(&sink, 64);
  hipMemset(buf, 1, total);
  const int iters = 2048;
  if (argc > 1 && argv[1][0] == 's') {      // store-side probe: how fast can one CU WRITE (the GEMM epilogue: 128 KiB per 256^2 tile)
    for (int nwg : {64, 256, 512}) {
      run<2, 8>("nt-store stream 2MiB/WG pitch 8K", buf, 8192, 256, nwg, sink, (long)2 << 20, 256);
      run<3, 8>("store    stream 2MiB/WG pitch 8K", buf, 8192, 256, nwg, sink, (long)2 << 20, 256);
      run<2, 4>("nt-store stream 2MiB/WG pitch 8K", buf, 8192, 256, nwg, sink, (long)2 << 20, 256);
      run<2, 8>("nt-store rows of 512 B (N=256 tile)", buf, 512, 4096, nwg, sink, (long)2 << 20, 256);
    }
    return 0;
  }
  for (int nwg : {160, 256, 512}) {
    // (a) tiny footprint: 256 rows x 128 B pitch = 32 KiB per WG (L1/L2 hits)
    run<0, 4>("dma  L2-resident 32KiB/WG", buf, 128, 256, nwg, sink, 65536, iters);
    run<1, 4>("vgpr L2-resident 32KiB/WG", buf, 128, 256, nwg, sink, 65536, iters);
    run<0, 8>("dma  L2-resident 32KiB/WG", buf, 128, 256, nwg, sink, 65536, iters);
    // (b) GEMM-like: 256 rows with an 8-KiB pitch (K = 4096 bf16), walking k: 2 MiB per WG, no reuse
    run<0, 4>("dma  stream 2MiB/WG pitch 8K", buf, 8192, 256, nwg, sink, (long)2 << 20, iters);
    run<1, 4>("vgpr stream 2MiB/WG pitch 8K", buf, 8192, 256, nwg, sink, (long)2 << 20, iters);
    run<0, 8>("dma  stream 2MiB/WG pitch 8K", buf, 8192, 256, nwg, sink, (long)2 << 20, iters);
    // (c) pitch 22016 B (K = 11008)
    run<0, 4>("dma  stream 5.4MiB/WG pitch 22K", buf, 22016, 256, nwg, sink, (long)5636096, iters);
    // (d) shared: every WG reads the SAME 2 MiB (all hit L2 / MALL after first touch)
    run<0, 4>("dma  shared 2MiB pitch 8K", buf, 8192, 256, nwg, sink, 0, iters);
  }
  return 0;
}
