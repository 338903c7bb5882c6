// preprocess.hip — crop + pad + Pillow-exact antialiased bicubic resize + CLIP normalisation on the GPU.
//
// Replaces the host side of VSM.inference (visual_search.py:186-194): for every crop
//   CLIP  : expand2square (TOP-LEFT paste on the CLIP-mean colour, VisualSearch/utils/utils.py:28-39) -> resize IxI
//   OWL   : resize (w x h) -> 768x768, aspect not preserved
// both with PIL.Image.resize(BICUBIC) on uint8 (Pillow Resample.c: two passes, window = 2*scale, coefficients quantised
// to 22 fractional bits, uint8 intermediate), then x/255, (x-mean)/std, bf16.  The full image stays resident in HBM;
// a crop is just a box.  Coefficient tables are computed on the host in the same double arithmetic as Pillow; the
// kernels do the integer convolution, so the uint8 result is bit-identical to Pillow and the bf16 output is a 256-entry
// lookup per channel.
#include "common.hpp"
#include "kernels.hpp"

namespace VS_NS {

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ int clip8(int v) {
  v >>= PRECISION_BITS;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// horizontal pass: temp[y][xo][c] for y in [0, in_h)
__global__ __launch_bounds__(256) void resize_h_kernel(const PreJob* __restrict__ jobs,
                                                       const int32_t* __restrict__ tables, uint8_t* __restrict__ temp,
                                                       int which) {
  const PreJob j = jobs[blockIdx.z * 2 + which];
  const int xo = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  if (xo >= j.out || y >= j.in_h) return;
  const int32_t* b = tables + j.hb_off + xo * 2;
  const int xmin = b[0], cnt = b[1];
  const int32_t* k = tables + j.hc_off + xo * j.hks;
  int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
  const bool row_in = y < j.ch;
  const uint8_t* row = j.img + ((int64_t)(j.y0 + (row_in ? y : 0)) * j.img_w + j.x0) * 3;      // the job's own image slot
  for (int t = 0; t < cnt; ++t) {
    const int sx = xmin + t;
    int r = 122, g = 116, bl = 104;                      // expand2square background = int(CLIP mean * 255)
    if (row_in && sx < j.cw) {
      const uint8_t* p = row + sx * 3;
      r = p[0]; g = p[1]; bl = p[2];
    }
    const int kk = k[t];
    s0 += r * kk; s1 += g * kk; s2 += bl * kk;
  }
  uint8_t* o = temp + j.temp_off + ((int64_t)y * j.out + xo) * 3;
  o[0] = (uint8_t)clip8(s0); o[1] = (uint8_t)clip8(s1); o[2] = (uint8_t)clip8(s2);
}

// vertical pass + normalisation LUT: out[c][yo][xo] bf16
__global__ __launch_bounds__(256) void resize_v_kernel(const PreJob* __restrict__ jobs, const int32_t* __restrict__ tables,
                                                       const uint8_t* __restrict__ temp, const lp_t* __restrict__ lut,
                                                       lp_t* __restrict__ out, int which) {
  const PreJob j = jobs[blockIdx.z * 2 + which];
  const int xo = blockIdx.x * blockDim.x + threadIdx.x;
  const int yo = blockIdx.y;
  if (xo >= j.out || yo >= j.out) return;
  const int32_t* b = tables + j.vb_off + yo * 2;
  const int ymin = b[0], cnt = b[1];
  const int32_t* k = tables + j.vc_off + yo * j.vks;
  int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
  const uint8_t* col = temp + j.temp_off + ((int64_t)ymin * j.out + xo) * 3;
  for (int t = 0; t < cnt; ++t) {
    const uint8_t* p = col + (int64_t)t * j.out * 3;
    const int kk = k[t];
    s0 += p[0] * kk; s1 += p[1] * kk; s2 += p[2] * kk;
  }
  const int64_t plane = (int64_t)j.out * j.out;
  lp_t* o = out + j.out_off + (int64_t)yo * j.out + xo;
  o[0] = lut[clip8(s0)];
  o[plane] = lut[256 + clip8(s1)];
  o[2 * plane] = lut[512 + clip8(s2)];
}

double bicubic(double x) {
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}

}  // namespace

// Pillow's precompute_coeffs + normalize_coeffs_8bpc for one axis.
void pil_bicubic_coeffs(int in_size, int out_size, std::vector<int32_t>* bounds, std::vector<int32_t>* coeffs, int* ksize_out) {
  const double scale = (double)in_size / out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 2.0 * filterscale;
  const int ksize = (int)ceil(support) * 2 + 1;
  bounds->assign((size_t)out_size * 2, 0);
  coeffs->assign((size_t)out_size * ksize, 0);
  std::vector<double> k((size_t)ksize);
  const double ss = 1.0 / filterscale;
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = (xx + 0.5) * scale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double ww = 0.0;
    for (int x = 0; x < ksize; ++x) k[x] = 0.0;
    for (int x = 0; x < xmax; ++x) {
      const double w = bicubic((x + xmin - center + 0.5) * ss);
      k[x] = w;
      ww += w;
    }
    for (int x = 0; x < xmax; ++x)
      if (ww != 0.0) k[x] /= ww;
    (*bounds)[(size_t)xx * 2] = xmin;
    (*bounds)[(size_t)xx * 2 + 1] = xmax;
    for (int x = 0; x < ksize; ++x) {
      const double v = k[x] * (1 << PRECISION_BITS);
      (*coeffs)[(size_t)xx * ksize + x] = k[x] < 0 ? (int)(-0.5 + v) : (int)(0.5 + v);
    }
  }
  *ksize_out = ksize;
}

// bf16((float(u8 * (1/255) in double) - mean) / std) per channel: HF rescale (double multiply, float32 cast) + normalize
void clip_norm_lut(lp_t* lut /*[3*256]*/) {
  const float mean[3] = {0.48145466f, 0.4578275f, 0.40821073f};
  const float stdv[3] = {0.26862954f, 0.26130258f, 0.27577711f};
  for (int c = 0; c < 3; ++c)
    for (int v = 0; v < 256; ++v) {
      const float r = (float)((double)v * 0.00392156862745098);
      lut[c * 256 + v] = f2lp((r - mean[c]) / stdv[c]);
    }
}

hipError_t preprocess_launch(const PreJob* jobs, const int32_t* tables, uint8_t* temp,
                             const lp_t* lut, lp_t* out, int which, int B, int out_size, int max_in_h, hipStream_t s) {
  dim3 gh((out_size + 255) / 256, max_in_h, B);
  hipLaunchKernelGGL(resize_h_kernel, gh, dim3(256), 0, s, jobs, tables, temp, which);
  dim3 gv((out_size + 255) / 256, out_size, B);
  hipLaunchKernelGGL(resize_v_kernel, gv, dim3(256), 0, s, jobs, tables, temp, lut, out, which);
  return hipGetLastError();
}

}  // namespace VS_NS
