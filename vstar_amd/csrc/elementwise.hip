// elementwise.hip — layout / gather / broadcast kernels of the VSM scoring path (HBM-bound, 16-byte accesses).
#include "common.hpp"
#include "kernels.hpp"

namespace VS_NS {

namespace {

// Conv2d(3, C, kernel=ps, stride=ps, bias=False) as a GEMM: A[b*P + py*G + px][c*ps*ps + ky*ps + kx]
// (HF CLIPVisionEmbeddings.patch_embedding / OwlViTVisionEmbeddings.patch_embedding; clip_encoder.py:53-57, owlvit.py:121-126)
__global__ void im2col_patch_kernel(const lp_t* __restrict__ pix, lp_t* __restrict__ A, int B, int I, int ps, int Kpad) {
  const int G = I / ps, P = G * G;
  const int K = 3 * ps * ps;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)B * P * Kpad;
  if (idx >= total) return;
  const int k = (int)(idx % Kpad);
  const int64_t row = idx / Kpad;
  lp_t v = 0;
  if (k < K) {
    const int p = (int)(row % P), b = (int)(row / P);
    const int c = k / (ps * ps), rem = k - c * ps * ps;
    const int ky = rem / ps, kx = rem - ky * ps;
    const int y = (p / G) * ps + ky, x = (p % G) * ps + kx;
    v = pix[(((int64_t)b * 3 + c) * I + y) * I + x];
  }
  A[idx] = v;
}

// embeddings = cat([class_embedding, patch_embeds]) + position_embedding   (bf16 add)
__global__ void vit_assemble_kernel(const lp_t* __restrict__ patch, const lp_t* __restrict__ cls,
                                    const lp_t* __restrict__ pos, lp_t* __restrict__ tokens, int B, int P, int C) {
  const int cv = C >> 3;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)B * (P + 1) * cv;
  if (idx >= total) return;
  const int v = (int)(idx % cv);
  const int64_t row = idx / cv;
  const int t = (int)(row % (P + 1)), b = (int)(row / (P + 1));
  const lpx8 a = (t == 0) ? *(const lpx8*)(cls + v * 8)
                            : *(const lpx8*)(patch + ((int64_t)b * P + (t - 1)) * C + v * 8);
  const lpx8 pe = *(const lpx8*)(pos + (int64_t)t * C + v * 8);
  lpx8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (short)f2lp(lp2f((lp_t)a[e]) + lp2f((lp_t)pe[e]));
  *(lpx8*)(tokens + row * C + v * 8) = o;
}

// prepare_inputs_labels_for_multimodal, mm_use_im_start_end branch (llava_arch.py:185-208,235-247):
// spliced position s < img_col -> ids[s]; s >= img_col+P -> ids[s-P+1]; the P image rows are left to the projector GEMM.
__global__ void llm_embed_text_kernel(const int32_t* __restrict__ ids, int L, int img_col, int P,
                                      const lp_t* __restrict__ table, int vocab, lp_t* __restrict__ x, int B, int C) {
  const int cv = C >> 3;
  const int T = L - 1;  // text tokens per row
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)B * T * cv;
  if (idx >= total) return;
  const int v = (int)(idx % cv);
  const int64_t r = idx / cv;
  const int t = (int)(r % T), b = (int)(r / T);
  const int col = t < img_col ? t : t + 1;          // column in ids
  const int s = t < img_col ? t : t + P;            // spliced position
  int id = ids[(int64_t)b * L + col];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  const int S = T + P;
  *(lpx8*)(x + ((int64_t)b * S + s) * C + v * 8) = *(const lpx8*)(table + (int64_t)id * C + v * 8);
}

__global__ void add_bcast_kernel(const lp_t* __restrict__ a, const lp_t* __restrict__ b, lp_t* __restrict__ out,
                                 int64_t rows, int cols, int64_t b_rows) {
  const int cv = cols >> 3;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * cv) return;
  const int v = (int)(idx % cv);
  const int64_t r = idx / cv;
  const lpx8 x = *(const lpx8*)(a + r * cols + v * 8);
  const lpx8 y = *(const lpx8*)(b + (r % b_rows) * cols + v * 8);
  lpx8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (short)f2lp(lp2f((lp_t)x[e]) + lp2f((lp_t)y[e]));
  *(lpx8*)(out + r * cols + v * 8) = o;
}

// grouped scoring: every crop's block of rows_per rows, plus the broadcast row b, repeated for the rep prompts of that crop
__global__ void bcast_rows_kernel(const lp_t* __restrict__ src, lp_t* __restrict__ dst, int nrep, int64_t rep_stride, int nrows, int cv,
                                  int64_t ld) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t per = (int64_t)nrows * cv;
  if (idx >= per * nrep) return;
  const int r = (int)(idx / per);
  const int64_t rem = idx - (int64_t)r * per;
  const int i = (int)(rem / cv), c = (int)(rem - (int64_t)i * cv);
  *(lpx8*)(dst + ((int64_t)r * rep_stride + i) * ld + c * 8) = *(const lpx8*)(src + (int64_t)i * ld + c * 8);
}

__global__ void add_bcast_repeat_kernel(const lp_t* __restrict__ a, const lp_t* __restrict__ b, lp_t* __restrict__ out, int n_out,
                                        int rep, int rows_per, int cols) {
  const int cv = cols >> 3;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)n_out * rows_per * cv) return;
  const int v = (int)(idx % cv);
  const int64_t r = idx / cv;
  const int n = (int)(r / rows_per), p = (int)(r % rows_per);
  const lpx8 x = *(const lpx8*)(a + ((int64_t)(n / rep) * rows_per + p) * cols + v * 8);
  const lpx8 y = *(const lpx8*)(b + v * 8);
  lpx8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (short)f2lp(lp2f((lp_t)x[e]) + lp2f((lp_t)y[e]));
  *(lpx8*)(out + r * cols + v * 8) = o;
}

// image_embeds[:, 1:, :] * image_embeds[:, :1, :]   (owlvit.py:131-137)
__global__ void owl_cls_mul_kernel(const lp_t* __restrict__ x, lp_t* __restrict__ y, int B, int N, int C) {
  const int cv = C >> 3;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)B * (N - 1) * cv;
  if (idx >= total) return;
  const int v = (int)(idx % cv);
  const int64_t r = idx / cv;
  const int p = (int)(r % (N - 1)), b = (int)(r / (N - 1));
  const lpx8 t = *(const lpx8*)(x + ((int64_t)b * N + 1 + p) * C + v * 8);
  const lpx8 c = *(const lpx8*)(x + ((int64_t)b * N) * C + v * 8);
  lpx8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (short)f2lp(lp2f((lp_t)t[e]) * lp2f((lp_t)c[e]));
  *(lpx8*)(y + r * C + v * 8) = o;
}

__global__ void gather_rows_kernel(const lp_t* __restrict__ x, const int32_t* __restrict__ idx_, lp_t* __restrict__ y,
                                   int rows, int cols) {
  const int cv = cols >> 3;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)rows * cv) return;
  const int v = (int)(idx % cv);
  const int r = (int)(idx / cv);
  *(lpx8*)(y + (int64_t)r * cols + v * 8) = *(const lpx8*)(x + (int64_t)idx_[r] * cols + v * 8);
}

// one block per row; first maximal index wins (torch.argmax tie rule on CPU/GPU is "first occurrence")
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ x, int cols, int ld, int32_t* out,
                                                          int out_stride) {
  __shared__ float sv[4];
  __shared__ int si[4];
  const int row = blockIdx.x;
  const float* xr = x + (int64_t)row * ld;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int c = threadIdx.x; c < cols; c += 256) {
    const float v = xr[c];
    if (v > best || (v == best && c < bi)) { best = v; bi = c; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w)
      if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
    out[(int64_t)row * out_stride] = bi;
  }
}

inline unsigned nblk(int64_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

hipError_t im2col_patch(const lp_t* pix, lp_t* A, int B, int I, int ps, int Kpad, hipStream_t s) {
  const int G = I / ps;
  const int64_t total = (int64_t)B * G * G * Kpad;
  hipLaunchKernelGGL(im2col_patch_kernel, dim3(nblk(total)), dim3(256), 0, s, pix, A, B, I, ps, Kpad);
  return hipGetLastError();
}
hipError_t vit_assemble_tokens(const lp_t* patch, const lp_t* cls, const lp_t* pos, lp_t* tokens, int B, int P,
                               int C, hipStream_t s) {
  if (C % 8) return hipErrorInvalidValue;
  hipLaunchKernelGGL(vit_assemble_kernel, dim3(nblk((int64_t)B * (P + 1) * (C / 8))), dim3(256), 0, s, patch, cls, pos,
                     tokens, B, P, C);
  return hipGetLastError();
}
hipError_t llm_embed_text(const int32_t* ids, int L, int img_col, int P, const lp_t* table, int vocab, lp_t* x, int B,
                          int C, hipStream_t s) {
  if (C % 8) return hipErrorInvalidValue;
  hipLaunchKernelGGL(llm_embed_text_kernel, dim3(nblk((int64_t)B * (L - 1) * (C / 8))), dim3(256), 0, s, ids, L, img_col,
                     P, table, vocab, x, B, C);
  return hipGetLastError();
}
hipError_t add_bcast(const lp_t* a, const lp_t* b, lp_t* out, int64_t rows, int cols, int64_t b_rows, hipStream_t s) {
  if (cols % 8) return hipErrorInvalidValue;
  hipLaunchKernelGGL(add_bcast_kernel, dim3(nblk(rows * (cols / 8))), dim3(256), 0, s, a, b, out, rows, cols, b_rows);
  return hipGetLastError();
}
hipError_t bcast_rows(const lp_t* src, lp_t* dst, int nrep, int64_t rep_stride, int nrows, int cols, int64_t ld, hipStream_t s) {
  if (cols % 8 || nrep <= 0 || nrows <= 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(bcast_rows_kernel, dim3(nblk((int64_t)nrep * nrows * (cols / 8))), dim3(256), 0, s, src, dst, nrep, rep_stride, nrows,
                     cols / 8, ld);
  return hipGetLastError();
}

hipError_t add_bcast_repeat(const lp_t* a, const lp_t* b, lp_t* out, int n_out, int rep, int rows_per, int cols, hipStream_t s) {
  if (cols % 8 || rep < 1) return hipErrorInvalidValue;
  hipLaunchKernelGGL(add_bcast_repeat_kernel, dim3(nblk((int64_t)n_out * rows_per * (cols / 8))), dim3(256), 0, s, a, b, out, n_out, rep,
                     rows_per, cols);
  return hipGetLastError();
}
hipError_t owl_cls_mul(const lp_t* x, lp_t* y, int B, int N, int C, hipStream_t s) {
  if (C % 8) return hipErrorInvalidValue;
  hipLaunchKernelGGL(owl_cls_mul_kernel, dim3(nblk((int64_t)B * (N - 1) * (C / 8))), dim3(256), 0, s, x, y, B, N, C);
  return hipGetLastError();
}
hipError_t gather_rows(const lp_t* x, const int32_t* idx, lp_t* y, int rows, int cols, hipStream_t s) {
  if (cols % 8) return hipErrorInvalidValue;
  if (rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(nblk((int64_t)rows * (cols / 8))), dim3(256), 0, s, x, idx, y, rows, cols);
  return hipGetLastError();
}
hipError_t argmax_rows(const float* x, int rows, int cols, int ld, int32_t* out, int out_stride, hipStream_t s) {
  if (rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(argmax_rows_kernel, dim3(rows), dim3(256), 0, s, x, cols, ld, out, out_stride);
  return hipGetLastError();
}

}  // namespace VS_NS
