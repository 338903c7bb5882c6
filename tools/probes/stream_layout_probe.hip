// stream_layout_probe.hip — does the LAYOUT of a weight matrix in HBM bound the decode GEMV's streaming rate?  (round 5)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/stream_layout_probe.hip -o tools/probes/stream_layout_probe.bin
// The decode GEMV (csrc/decode.hip::gemm_skinny_ring_kernel) gives every workgroup 16 (or 32) ROWS of a row-major [N][K] matrix:
// its eight waves walk K in 128-byte steps, one LDS-DMA request = 8 rows x 128 B, so a workgroup keeps 16 - 32 separate
// sequential streams open, the chip 4096 - 11008 of them.  This probe streams the same bytes with the same request shape, ring depth
// and grid, once in that ROW layout and once TILED (the 1-KiB pieces of a workgroup laid out back to back: one sequential stream per
// workgroup), and prints GB/s.  If the tiled form is much faster, a second, tile-major copy of the weights for the decode path
// (HBM has room) would lift the 4.3 TB/s the GEMVs stream at.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// rows_per_wg rows of K bytes each (row pitch = kbytes); wave w takes 128-byte k-steps w, w + 8, ...; a request covers 8 rows.
template <bool TILED>
__global__ __launch_bounds__(512, 2) void stream(const char* src, int rows_per_wg, long kbytes, int* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int st_r = lane >> 3, st_c = lane & 7;
  char* ring = smem + wave * 10240;
  const int groups = rows_per_wg / 8;                          // requests per k-step
  const long nsteps = kbytes / 128;
  const long wg_bytes = (long)rows_per_wg * kbytes;
  const char* base = src + (long)blockIdx.x * wg_bytes;
  constexpr int RD = 3;
  long issued = 0;
  auto issue = [&](long i) {                                   // i-th k-step of this wave
    const long ks = wave + i * 8;
    char* st = ring + (i % RD) * 3072;
    for (int g = 0; g < groups; ++g) {
      const char* p = TILED ? base + (ks * groups + g) * 1024 + lane * 16
                            : base + (long)(g * 8 + st_r) * kbytes + ks * 128 + st_c * 16;
      __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(st + g * 1024), 16, 0, 0);
    }
  };
  const long n = (nsteps - wave + 7) / 8;
  for (long j = 0; j < RD && j < n; ++j) issue(j);
  for (long i = 0; i < n; ++i) {
    // wait for step i (the oldest): RD - 1 younger steps may stay in flight
    if (groups == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    if (i + RD < n) issue(i + RD);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (smem[threadIdx.x] == 77 && kbytes < 0) sink[0] = 1;
}

template <bool TILED>
double run(const char* buf, int nwg, int rows_per_wg, long kbytes, int* sink, int iters) {
  auto k = stream<TILED>;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k, dim3(nwg), dim3(512), 81920, 0, buf, rows_per_wg, kbytes, sink);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int w = 0; w < iters; ++w) hipLaunchKernelGGL(k, dim3(nwg), dim3(512), 81920, 0, buf, rows_per_wg, kbytes, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  return (double)nwg * rows_per_wg * kbytes / (ms / iters * 1e-3) / 1e9;
}

int main() {
  // rotate over several copies so that nothing is served from the 256-MiB Infinity Cache
  const long copy = 512l << 20;
  char* buf; hipMalloc(&buf, copy * 3); hipMemset(buf, 1, copy * 3);
  int* sink; hipMalloc(&sink, 16);
  struct { const char* name; int n; long k; int rows; } shapes[] = {
      {"o_proj    N 4096  K 4096 ", 4096, 4096, 16}, {"down_proj N 4096  K 11008", 4096, 11008, 16}, {"qkv       N 12288 K 4096 ", 12288, 4096, 16},
      {"gate|up   N 22016 K 4096 ", 22016, 4096, 32}};
  printf("%-28s %10s %10s   (GB/s; one launch streams the matrix once; 8 waves x 3-deep LDS-DMA ring per workgroup, 2 workgroups per CU)\n", "shape", "row-major", "tiled");
  for (auto& s : shapes) {
    const long kb = s.k * 2;
    const int nwg = s.n / s.rows;
    double a = 0, b = 0;
    for (int rep = 0; rep < 3; ++rep) {
      const char* p = buf + (rep % 3) * copy;
      a = run<false>(p, nwg, s.rows, kb, sink, 20) > a ? run<false>(p, nwg, s.rows, kb, sink, 20) : a;
      b = run<true>(p, nwg, s.rows, kb, sink, 20) > b ? run<true>(p, nwg, s.rows, kb, sink, 20) : b;
    }
    printf("%-28s %10.0f %10.0f   x%.3f\n", s.name, a, b, b / a);
  }
  return 0;
}
