// heads.hip — OWL-ViT class/box head finishers and the SAM-style mask head's non-GEMM pieces.
#include "common.hpp"
#include "kernels.hpp"

namespace VS_NS {

namespace {

// HF OwlViTClassPredictionHead.forward with ONE query per crop (owlvit.py:102-119,150-170):
//   e = dense0(x); e /= (||e|| + 1e-6); q /= (||q|| + 1e-6); logit = (e.q + shift(x)) * (elu(scale(x)) + 1)
// emb row layout (fp32, from one fused GEMM): [0,Q) dense0 | Q shift | Q+1 scale.  bf16 rounding points as in the
// bf16 reference.  One wave per image token.
__global__ __launch_bounds__(256) void owl_class_kernel(const float* __restrict__ emb, int ld, int Q,
                                                        const lp_t* __restrict__ query, float* __restrict__ out,
                                                        int out_stride_crop, int B, int rows_per_crop, int img_div) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (int64_t)B * rows_per_crop) return;
  const int b = (int)(row / rows_per_crop), p = (int)(row % rows_per_crop);
  const float* er = emb + ((int64_t)(b / img_div) * rows_per_crop + p) * ld;
  const lp_t* qr = query + (int64_t)b * Q;
  float ee = 0.f, qq = 0.f;
  for (int d = lane; d < Q; d += 64) {
    const float e = rlp(er[d]), q = lp2f(qr[d]);
    ee += e * e;
    qq += q * q;
  }
  ee = wave_sum(ee);
  qq = wave_sum(qq);
  const float en = rlp(rlp(sqrtf(ee)) + 1e-6f), qn = rlp(rlp(sqrtf(qq)) + 1e-6f);
  float dot = 0.f;
  for (int d = lane; d < Q; d += 64) dot += rlp(rlp(er[d]) / en) * rlp(lp2f(qr[d]) / qn);
  dot = rlp(wave_sum(dot));
  if (lane == 0) {
    const float shift = rlp(er[Q]);
    const float sc = rlp(er[Q + 1]);
    const float elu = sc > 0.f ? sc : (__expf(sc) - 1.0f);
    const float scale = rlp(rlp(elu) + 1.0f);
    out[(int64_t)b * out_stride_crop + p] = rlp(rlp(dot + shift) * scale);
  }
}

// box_predictor: pred = sigmoid(box_head(x) + box_bias)  (owlvit.py:63-100)
__global__ void owl_box_kernel(const float* __restrict__ raw, int ld, float* __restrict__ out, int out_stride_crop, int B,
                               int grid, int img_div) {
  const int npatch = grid * grid;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)B * npatch * 4) return;
  const int c = (int)(idx & 3);
  const int64_t row = idx >> 2;
  const int p = (int)(row % npatch), b = (int)(row / npatch);
  float coord;
  if (c == 0) coord = (float)(p % grid + 1) / (float)grid;
  else if (c == 1) coord = (float)(p / grid + 1) / (float)grid;
  else coord = 1.0f / (float)grid;
  coord = fminf(fmaxf(coord, 0.f), 1.f);
  const float bias = logf(coord + 1e-4f) - log1pf(-coord + 1e-4f);
  const float v = rlp(rlp(raw[((int64_t)(b / img_div) * npatch + p) * ld + c]) + bias);
  out[(int64_t)b * out_stride_crop + p * 4 + c] = rlp(1.0f / (1.0f + __expf(-v)));
}

// mask_decoder.Upsample: F.interpolate(x.float(), scale_factor=2, "bilinear").to(bf16) then Conv2d 3x3 pad 1
// (mask_decoder.py:15-27).  Emits the conv's im2col matrix directly: A[(b,Y,X)][(ky*3+kx)*C + c].
__global__ void up2x_im2col_kernel(const lp_t* __restrict__ src, lp_t* __restrict__ A, int B, int h, int w, int C) {
  const int cv = C >> 3;
  const int H2 = 2 * h, W2 = 2 * w;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)B * H2 * W2 * 9 * cv;
  if (idx >= total) return;
  const int v = (int)(idx % cv);
  int64_t r = idx / cv;
  const int tap = (int)(r % 9);
  r /= 9;
  const int X = (int)(r % W2);
  const int Y = (int)((r / W2) % H2);
  const int b = (int)(r / ((int64_t)W2 * H2));
  const int yy = Y + tap / 3 - 1, xx = X + tap % 3 - 1;
  lpx8 o = {0, 0, 0, 0, 0, 0, 0, 0};
  if (yy >= 0 && yy < H2 && xx >= 0 && xx < W2) {
    const float sy = fmaxf(0.5f * (yy + 0.5f) - 0.5f, 0.f), sx = fmaxf(0.5f * (xx + 0.5f) - 0.5f, 0.f);
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
    const float ly = sy - y0, lx = sx - x0;
    const lp_t* base = src + (int64_t)b * h * w * C + v * 8;
    const lpx8 v00 = *(const lpx8*)(base + ((int64_t)y0 * w + x0) * C);
    const lpx8 v01 = *(const lpx8*)(base + ((int64_t)y0 * w + x1) * C);
    const lpx8 v10 = *(const lpx8*)(base + ((int64_t)y1 * w + x0) * C);
    const lpx8 v11 = *(const lpx8*)(base + ((int64_t)y1 * w + x1) * C);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float t = (1.f - ly) * ((1.f - lx) * lp2f((lp_t)v00[e]) + lx * lp2f((lp_t)v01[e])) +
                      ly * ((1.f - lx) * lp2f((lp_t)v10[e]) + lx * lp2f((lp_t)v11[e]));
      o[e] = (short)f2lp(t);
    }
  }
  *(lpx8*)(A + idx * 8) = o;
}

// masks = hyper_in @ upscaled_embedding.view(b, c, h*w)  (mask_decoder.py:176-181), mask token 0 only
template <int C>
__global__ void hyper_mask_kernel(const lp_t* __restrict__ hyper, const lp_t* __restrict__ up, float* __restrict__ out,
                                  int out_stride_crop, int B, int npix) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)B * npix) return;
  const int b = (int)(idx / npix), pix = (int)(idx % npix);
  const lp_t* u = up + idx * C;
  const lp_t* hy = hyper + (int64_t)b * C;
  float a = 0.f;
#pragma unroll
  for (int v = 0; v < C / 8; ++v) {
    const lpx8 x = *(const lpx8*)(u + v * 8);
    const lpx8 w = *(const lpx8*)(hy + v * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) a += lp2f((lp_t)x[e]) * lp2f((lp_t)w[e]);
  }
  out[(int64_t)b * out_stride_crop + pix] = rlp(a);
}

// F.interpolate(low_res.float(), (h, w), "bilinear", align_corners=False) + clamp(min=0)  (VSM.py:534-537, visual_search.py:223-224)
__global__ void resize_bilinear_kernel(const float* __restrict__ in, int hin, int win, float* __restrict__ out, int hout,
                                       int wout, float rh, float rw, int clamp_min0) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)hout * wout) return;
  const int x = (int)(idx % wout), y = (int)(idx / wout);
  const float sy = fmaxf(rh * (y + 0.5f) - 0.5f, 0.f), sx = fmaxf(rw * (x + 0.5f) - 0.5f, 0.f);
  const int y0 = (int)sy, x0 = (int)sx;
  const int y1 = y0 + (y0 < hin - 1 ? 1 : 0), x1 = x0 + (x0 < win - 1 ? 1 : 0);
  const float ly = sy - y0, lx = sx - x0;
  const float t = (1.f - ly) * ((1.f - lx) * in[y0 * win + x0] + lx * in[y0 * win + x1]) +
                  ly * ((1.f - lx) * in[y1 * win + x0] + lx * in[y1 * win + x1]);
  out[idx] = clamp_min0 ? fmaxf(t, 0.f) : t;
}

// Decision statistics of one heat map WITHOUT materialising it (visual_search.py:255-275,420-426):
// H = clamp(bilinear(low_res -> hout x wout), 0); out = {min H, max H, sum H, sum H over each of n_rects rectangles}.
// Same interpolation code as resize_bilinear_kernel; fp64 accumulation.
constexpr int MAX_RECTS = 8;
// One launch for n maps (blockIdx.y = map): every map has its own output size and rectangles.  Per map `out` = 3 + MAX_RECTS doubles,
// `mm` = 2 words (min / max as uint: H >= 0, so uint order == float order), `rects` = MAX_RECTS x 4 ints, `hw` = (hout, wout).
// Block-level reduction through LDS, then ONE atomic per block and quantity (round 3: one per wave, and three launches per map —
// a search step of 32 searches queued ~200 small launches whose atomics serialised on eleven addresses).
__global__ __launch_bounds__(256) void heat_stats_kernel(const float* __restrict__ in_all, int hin, int win, const int* __restrict__ hw,
                                                         const int* __restrict__ rects_all, const int* __restrict__ n_rects_all,
                                                         double* __restrict__ out_all, unsigned* __restrict__ mm_all) {
  const int item = blockIdx.y;
  const float* in = in_all + (size_t)item * hin * win;
  const int hout = hw[2 * item], wout = hw[2 * item + 1], n_rects = n_rects_all[item];
  const int* rects = rects_all + item * MAX_RECTS * 4;
  double* out = out_all + item * (3 + MAX_RECTS);
  unsigned* mm = mm_all + item * 2;
  const float rh = (float)hin / (float)hout, rw = (float)win / (float)wout;
  const int64_t total = (int64_t)hout * wout;
  float mn = INFINITY, mx = 0.f;
  double sum = 0.0, rs[MAX_RECTS];
#pragma unroll
  for (int k = 0; k < MAX_RECTS; ++k) rs[k] = 0.0;
  int rx[MAX_RECTS], ry[MAX_RECTS], rx1[MAX_RECTS], ry1[MAX_RECTS];
#pragma unroll
  for (int k = 0; k < MAX_RECTS; ++k) {
    const bool on = k < n_rects;
    rx[k] = on ? rects[k * 4] : 0; ry[k] = on ? rects[k * 4 + 1] : 0;
    rx1[k] = on ? rx[k] + rects[k * 4 + 2] : 0; ry1[k] = on ? ry[k] + rects[k * 4 + 3] : 0;
  }
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(idx % wout), y = (int)(idx / wout);
    const float sy = fmaxf(rh * (y + 0.5f) - 0.5f, 0.f), sx = fmaxf(rw * (x + 0.5f) - 0.5f, 0.f);
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < hin - 1 ? 1 : 0), x1 = x0 + (x0 < win - 1 ? 1 : 0);
    const float ly = sy - y0, lx = sx - x0;
    const float t = (1.f - ly) * ((1.f - lx) * in[y0 * win + x0] + lx * in[y0 * win + x1]) +
                    ly * ((1.f - lx) * in[y1 * win + x0] + lx * in[y1 * win + x1]);
    const float hv = fmaxf(t, 0.f);
    mn = fminf(mn, hv);
    mx = fmaxf(mx, hv);
    sum += hv;
#pragma unroll
    for (int k = 0; k < MAX_RECTS; ++k)
      if (x >= rx[k] && x < rx1[k] && y >= ry[k] && y < ry1[k]) rs[k] += hv;
  }
  // wave reduce -> LDS -> one atomic per block
  for (int o = 32; o > 0; o >>= 1) {
    mn = fminf(mn, __shfl_xor(mn, o, 64));
    mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    sum += __shfl_xor(sum, o, 64);
#pragma unroll
    for (int k = 0; k < MAX_RECTS; ++k) rs[k] += __shfl_xor(rs[k], o, 64);
  }
  __shared__ double s_sum[4][1 + MAX_RECTS];
  __shared__ float s_mn[4], s_mx[4];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    s_mn[wave] = mn; s_mx[wave] = mx; s_sum[wave][0] = sum;
#pragma unroll
    for (int k = 0; k < MAX_RECTS; ++k) s_sum[wave][1 + k] = rs[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicMin(&mm[0], __float_as_uint(fminf(fminf(s_mn[0], s_mn[1]), fminf(s_mn[2], s_mn[3]))));
    atomicMax(&mm[1], __float_as_uint(fmaxf(fmaxf(s_mx[0], s_mx[1]), fmaxf(s_mx[2], s_mx[3]))));
  }
  if (threadIdx.x <= n_rects)
    atomicAdd(&out[2 + threadIdx.x], (s_sum[0][threadIdx.x] + s_sum[1][threadIdx.x]) + (s_sum[2][threadIdx.x] + s_sum[3][threadIdx.x]));
}
__global__ void heat_stats_init(double* out, unsigned* mm, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n * (3 + MAX_RECTS)) out[i] = 0.0;
  if (i < n) { mm[2 * i] = 0x7f800000u; mm[2 * i + 1] = 0u; }
}
__global__ void heat_stats_finish(double* out, const unsigned* mm, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i * (3 + MAX_RECTS)] = (double)__uint_as_float(mm[2 * i]);
  out[i * (3 + MAX_RECTS) + 1] = (double)__uint_as_float(mm[2 * i + 1]);
}

inline unsigned nblk(int64_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

// n maps in one launch.  Device arrays: lowres [n][hin*win], hw [n][2] = (hout, wout), rects [n][MAX_RECTS*4], n_rects [n],
// out [n][3+MAX_RECTS], mm [n][2].  max_pixels = the largest hout*wout among the maps (sizes the grid).
hipError_t heat_stats_batch(const float* lowres, int hin, int win, const int* hw, const int* rects, const int* n_rects, int n,
                            int64_t max_pixels, double* out, unsigned* mm_scratch, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  const int cells = n * (3 + MAX_RECTS);
  hipLaunchKernelGGL(heat_stats_init, dim3((cells + 255) / 256), dim3(256), 0, s, out, mm_scratch, n);
  // >= 8 pixels per thread, at most 256 blocks per map (one block per CU for a 4K map; the small maps of deep nodes take fewer)
  int64_t bx = (max_pixels + 256 * 8 - 1) / (256 * 8);
  if (bx < 1) bx = 1;
  if (bx > 256) bx = 256;
  hipLaunchKernelGGL(heat_stats_kernel, dim3((unsigned)bx, (unsigned)n), dim3(256), 0, s, lowres, hin, win, hw, rects, n_rects, out,
                     mm_scratch);
  hipLaunchKernelGGL(heat_stats_finish, dim3((n + 255) / 256), dim3(256), 0, s, out, mm_scratch, n);
  return hipGetLastError();
}

hipError_t owl_class_logits(const float* emb, int ld, int Q, const lp_t* query, float* out, int out_stride_crop, int B,
                            int rows_per_crop, hipStream_t s, int img_div) {
  const int64_t rows = (int64_t)B * rows_per_crop;
  hipLaunchKernelGGL(owl_class_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, emb, ld, Q, query, out,
                     out_stride_crop, B, rows_per_crop, img_div < 1 ? 1 : img_div);
  return hipGetLastError();
}
hipError_t owl_box_finish(const float* raw, int ld, float* out, int out_stride_crop, int B, int grid, hipStream_t s, int img_div) {
  hipLaunchKernelGGL(owl_box_kernel, dim3(nblk((int64_t)B * grid * grid * 4)), dim3(256), 0, s, raw, ld, out,
                     out_stride_crop, B, grid, img_div < 1 ? 1 : img_div);
  return hipGetLastError();
}
hipError_t upsample2x_im2col3x3(const lp_t* src, lp_t* A, int B, int h, int w, int C, hipStream_t s) {
  if (C % 8) return hipErrorInvalidValue;
  hipLaunchKernelGGL(up2x_im2col_kernel, dim3(nblk((int64_t)B * 4 * h * w * 9 * (C / 8))), dim3(256), 0, s, src, A, B, h,
                     w, C);
  return hipGetLastError();
}
hipError_t hyper_mask(const lp_t* hyper, const lp_t* up, float* out, int out_stride_crop, int B, int npix, int C,
                      hipStream_t s) {
  if (C != 32) return hipErrorInvalidValue;
  hipLaunchKernelGGL(hyper_mask_kernel<32>, dim3(nblk((int64_t)B * npix)), dim3(256), 0, s, hyper, up, out,
                     out_stride_crop, B, npix);
  return hipGetLastError();
}
hipError_t resize_bilinear_clamp(const float* in, int hin, int win, float* out, int hout, int wout, hipStream_t s, int clamp_min0) {
  hipLaunchKernelGGL(resize_bilinear_kernel, dim3(nblk((int64_t)hout * wout)), dim3(256), 0, s, in, hin, win, out, hout,
                     wout, (float)hin / (float)hout, (float)win / (float)wout, clamp_min0);
  return hipGetLastError();
}

}  // namespace VS_NS
