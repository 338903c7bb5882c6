// store_pattern_probe.hip — what does a 256^2 GEMM tile's output store cost per CU, and does the LANE -> ADDRESS pattern matter?  (round 6)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/store_pattern_probe.hip -o tools/probes/store_pattern_probe.bin
// gemm4w's epilogue spends 3 - 4 us per tile on store-ISSUE back-pressure (tools/gemm4w_timeline.py): 32 global_store_dwordx4 per wave,
// 128 KiB per CU, every CU at once.  In the MFMA accumulator layout lane (fr = l & 15, fq = l >> 4) holds 16 bytes of ROW fr: one
// instruction writes 16 rows x 64 B, and consecutive LANES hit different rows (8 - 24 KB apart).  This probe writes the same bytes per CU
// (`tiles` x 128 KiB, 4 waves, all CUs at once, nothing else running) in four lane -> address patterns and prints us per tile and CU:
//   0 mfma     : lane -> row (l & 15), 16-byte chunk (l >> 4); second value of the lane at +64 B (the shipped epilogue)
//   1 rows8x128: lane -> row (l >> 3), chunk (l & 7): 8 rows x one whole 128-byte line per instruction (what an LDS transpose would give)
//   2 linear   : lane -> 16 B x l of one contiguous 1-KiB block per instruction (upper bound of the store path)
//   3 pairs    : the lane-pair re-deal of round 6's full-line experiment (even / odd fr share a row: 8 lanes x 16 B per line, lanes not adjacent)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) int i32x4;

template <int PAT>
__global__ __launch_bounds__(256, 1) void store_tiles(char* out, long ld, int tiles, int nt) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  i32x4 v = {lane, wave, (int)blockIdx.x, 7};
  for (int t = 0; t < tiles; ++t) {
    // tile (blockIdx, t): 256 rows x 512 B, wave (wr, wc) owns rows wr*128.., bytes wc*256..; row pitch ld
    char* tile = out + ((long)(t * gridDim.x + blockIdx.x) * 256) * ld;
    char* wbase = tile + (long)(wr * 128) * ld + wc * 256;
#pragma unroll
    for (int m = 0; m < 8; ++m) {            // 8 row groups of 16 rows, 4 x 16 B per lane and row group = 32 stores per wave and tile
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        char* p;
        if (PAT == 0) p = wbase + (long)(m * 16 + (lane & 15)) * ld + q * 64 + (lane >> 4) * 16;
        else if (PAT == 1) p = wbase + (long)(m * 16 + q * 4 + ((lane >> 3) & 3) + (lane >> 5) * 0) * ld + ((lane >> 5) * 128) + (lane & 7) * 16;
        else if (PAT == 2) p = wbase + (long)(m * 16 + q * 4 + (lane >> 4)) * ld + (lane & 15) * 16;
        else p = wbase + (long)(m * 16 + ((lane & 15) & ~1) + (q & 1)) * ld + (q >> 1) * 128 + ((lane & 1) * 64) + (lane >> 4) * 16;
        v[0] += 1;
        if (nt) __builtin_nontemporal_store(v, (i32x4*)p);
        else *(i32x4*)p = v;
      }
    }
  }
}

template <int PAT>
float run(char* out, long ld, int tiles, int nt, int cus) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  store_tiles<PAT><<<cus, 256>>>(out, ld, tiles, nt);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) store_tiles<PAT><<<cus, 256>>>(out, ld, tiles, nt);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}

int main() {
  const int cus = 256, tiles = 16;
  const char* names[4] = {"mfma (16 rows x 64 B)", "rows8x128 (8 lines)", "linear 1 KiB", "lane pairs (8 lines)"};
  for (long ld : {8192L, 22016L}) {
    char* out;
    const size_t bytes = (size_t)tiles * cus * 256 * ld;
    if (hipMalloc(&out, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    for (int nt = 0; nt < 2; ++nt) {
      float ms[4] = {run<0>(out, ld, tiles, nt, cus), run<1>(out, ld, tiles, nt, cus), run<2>(out, ld, tiles, nt, cus), run<3>(out, ld, tiles, nt, cus)};
      for (int p = 0; p < 4; ++p)
        printf("row pitch %6ld B  %s  %-24s %7.2f us per tile and CU   %6.1f GB/s per CU   %5.2f TB/s chip\n", ld, nt ? "nontemporal" : "plain      ", names[p],
               ms[p] * 1e3 / tiles, 131072.0 / (ms[p] * 1e-3 / tiles) / 1e9, 131072.0 * cus * tiles / (ms[p] * 1e-3) / 1e12);
    }
    hipFree(out);
  }
  return 0;
}
