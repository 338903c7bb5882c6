// mfma_scale_probe.hip — which lane's E8M0 byte scales which operand elements of v_mfma_scale_f32_16x16x128_f8f6f4 (gfx950), and how
// op_sel / op_sel_hi pick the byte of the scale register.  Evidence for the MX (block-scaled) W8A8 path of gemm4w.hip.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_scale_probe.hip -o /tmp/mfma_scale_probe && /tmp/mfma_scale_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int OA, int OB>
__global__ void k(const i32x8* a, const i32x8* b, f32x4* d, const int* sa, const int* sb) {
  const int l = threadIdx.x;
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], c, 0, 0, OA, sa[l], OB, sb[l]);
  d[l] = c;
}

template <int OA, int OB>
static std::vector<float> run(const std::vector<int>& sa, const std::vector<int>& sb, const std::vector<int>& av, const std::vector<int>& bv) {
  int *da, *db, *dsa, *dsb;
  float* dd;
  hipMalloc(&da, 64 * 32); hipMalloc(&db, 64 * 32); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dd, 64 * 16);
  hipMemcpy(da, av.data(), 64 * 32, hipMemcpyHostToDevice); hipMemcpy(db, bv.data(), 64 * 32, hipMemcpyHostToDevice);
  hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL((k<OA, OB>), dim3(1), dim3(64), 0, 0, (const i32x8*)da, (const i32x8*)db, (f32x4*)dd, dsa, dsb);
  std::vector<float> out(256);
  hipMemcpy(out.data(), dd, 1024, hipMemcpyDeviceToHost);
  hipFree(da); hipFree(db); hipFree(dsa); hipFree(dsb); hipFree(dd);
  return out;      // out[l * 4 + e] = D[4 * (l / 16) + e][l % 16]
}

int main() {
  std::vector<int> ones(64 * 8, 0x38383838), s1(64, 0x7f7f7f7f), s(64);
  // T1: scale of src1 = 2^(lane % 16): D[i][j] = 128 * 2^j if the byte of lane l scales the 32 elements lane l holds (column j = l % 16)
  for (int l = 0; l < 64; ++l) s[l] = 0x7f7f7f00 | (127 + l % 16);
  auto d = run<0, 0>(s1, s, ones, ones);
  printf("T1 src1 scale 2^(lane%%16): D[0][j], j = 0..15:");
  for (int j = 0; j < 16; ++j) printf(" %g", d[j * 4]);
  printf("\n   D[i][3], i = 0..15:");
  for (int i = 0; i < 16; ++i) printf(" %g", d[((i / 4) * 16 + 3) * 4 + i % 4]);
  // T2: scale of src1 = 2^(lane / 16): D = 32 * (1 + 2 + 4 + 8) = 480 everywhere if lane l's byte goes with k-block l / 16
  for (int l = 0; l < 64; ++l) s[l] = 0x7f7f7f00 | (127 + l / 16);
  d = run<0, 0>(s1, s, ones, ones);
  printf("\nT2 src1 scale 2^(lane/16): D[0][0] %g D[5][9] %g D[15][15] %g (480 = every k-block of 32 has its own lane group's scale)\n", d[0], d[(16 + 9) * 4 + 1], d[(48 + 15) * 4 + 3]);
  // T2b: only k-block 2 of src1 non-zero, scale 2^(lane/16): D = 32 * 4
  {
    std::vector<int> bz(64 * 8, 0);
    for (int l = 32; l < 48; ++l) for (int r = 0; r < 8; ++r) bz[l * 8 + r] = 0x38383838;
    d = run<0, 0>(s1, s, ones, bz);
    printf("T2b src1 non-zero only in lanes 32..47 (k-block 2), scale 2^(lane/16): D[0][0] %g (128 = 32 * 2^2)\n", d[0]);
  }
  // T3: scale of src0 = 2^(lane % 16): D[i][j] = 128 * 2^i
  for (int l = 0; l < 64; ++l) s[l] = 0x7f7f7f00 | (127 + l % 16);
  d = run<0, 0>(s, s1, ones, ones);
  printf("T3 src0 scale 2^(lane%%16): D[i][0], i = 0..15:");
  for (int i = 0; i < 16; ++i) printf(" %g", d[((i / 4) * 16) * 4 + i % 4]);
  // T4: byte select
  for (int l = 0; l < 64; ++l) s[l] = 0x7f | 0x80 << 8 | 0x81 << 16 | 0x82 << 24;
  printf("\nT4 src1 scale bytes {2^0, 2^1, 2^2, 2^3}: opsel 0..3 ->");
  printf(" %g", run<0, 0>(s1, s, ones, ones)[0]);
  printf(" %g", run<0, 1>(s1, s, ones, ones)[0]);
  printf(" %g", run<0, 2>(s1, s, ones, ones)[0]);
  printf(" %g", run<0, 3>(s1, s, ones, ones)[0]);
  printf("   (src0, opsel 0..3:");
  printf(" %g", run<0, 0>(s, s1, ones, ones)[0]);
  printf(" %g", run<1, 0>(s, s1, ones, ones)[0]);
  printf(" %g", run<2, 0>(s, s1, ones, ones)[0]);
  printf(" %g)\n", run<3, 0>(s, s1, ones, ones)[0]);
  return 0;
}
