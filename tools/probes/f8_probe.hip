// Probe: operand layout of v_mfma_scale_f32_16x16x128_f8f6f4 (fp8 e4m3, scales = 1).  Every lane holds 32 bytes per operand;
// hypotheses for which k those bytes are:  H0: k = 32*(lane/16) + b     H1: k = 16*(lane/16) + (b%16) + 64*(b/16)
// H2: k = 8*(lane/16) + (b%8) + 32*(b/8).   Prints the max |C - ref| under each.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void k(const i32x8* a, const i32x8* b, f32x4* c) {
  f32x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  c[threadIdx.x] = acc;
}
static float e4m3(uint8_t v) {   // OCP e4m3fn
  int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float f = e == 0 ? ldexpf((float)m / 8.f, -6) : ldexpf(1.f + m / 8.f, e - 7);
  return s ? -f : f;
}
int main() {
  uint8_t ha[64 * 32], hb[64 * 32];
  srand(1);
  for (int i = 0; i < 64 * 32; ++i) { ha[i] = (rand() % 0x38) | ((rand() & 1) << 7); hb[i] = (rand() % 0x38) | ((rand() & 1) << 7); }
  void *da, *db, *dc;
  hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dc, 64 * 16);
  hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, (const i32x8*)da, (const i32x8*)db, (f32x4*)dc);
  float hc[64 * 4];
  hipMemcpy(hc, dc, sizeof(hc), hipMemcpyDeviceToHost);
  for (int hyp = 0; hyp < 3; ++hyp) {
    static float A[16][128], B[16][128];
    for (int l = 0; l < 64; ++l)
      for (int bb = 0; bb < 32; ++bb) {
        int g = l / 16, kk;
        if (hyp == 0) kk = 32 * g + bb;
        else if (hyp == 1) kk = 16 * g + (bb % 16) + 64 * (bb / 16);
        else kk = 8 * g + (bb % 8) + 32 * (bb / 8);
        A[l % 16][kk] = e4m3(ha[l * 32 + bb]);
        B[l % 16][kk] = e4m3(hb[l * 32 + bb]);
      }
    // C/D: col = lane & 15, row = (lane >> 4) * 4 + reg  with row <- A's index, col <- B's index
    double err = 0, errT = 0;
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < 4; ++r) {
        int row = (l >> 4) * 4 + r, col = l & 15;
        double ref = 0, refT = 0;
        for (int kk = 0; kk < 128; ++kk) { ref += (double)A[row][kk] * B[col][kk]; refT += (double)A[col][kk] * B[row][kk]; }
        err = fmax(err, fabs(ref - hc[l * 4 + r])); errT = fmax(errT, fabs(refT - hc[l * 4 + r]));
      }
    printf("hypothesis %d: max err %.4g (A rows -> C rows), %.4g (transposed)\n", hyp, err, errT);
  }
  return 0;
}
