// kernels.hpp — host-side launch interface of the gfx950 kernels (internal; the public ABI is include/vstar_hip.h)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>
#include "../../include/vstar_hip.h"
#include "common.hpp"

namespace VS_NS {

// Row maps let a GEMM read/write a strided sub-sequence without a gather pass:
//   mapped(r) = (r / group) * gstride + off + r % group      (group <= 0: identity)
struct GemmParams {
  const lp_t* A; int64_t lda; int a_group; int64_t a_gstride; int64_t a_off;
  const lp_t* W;          // [ceil(N/256)*256, K] K-contiguous rows (nn.Linear layout), zero padded; K % 64 == 0
  const lp_t* bias;       // [N] or null
  const lp_t* res; int64_t ldr;   // residual, laid out like C (same row map), or null
  void* C; int64_t ldc; int c_group; int64_t c_gstride; int64_t c_off;
  int M, N, K;
  int debug_flags;   // diagnostics only: bit0 = skip the epilogue's global stores, bit1 = skip the whole epilogue, bit2 = gemm256 without its direct (in-register) epilogue
  // gemm_skinny_lp only: when norm_w != null the rows of A are RMS-normalised on the fly (LlamaRMSNorm: fp32 statistics,
  // rlp(x * rstd), then rlp(norm_w * that)) — bit-identical to rmsnorm_lp followed by the GEMM, one launch less
  const lp_t* norm_w; float norm_eps;
  // gemm256 only (bf16/fp16 output, VSTAR_EPI_NONE): rotate-half RoPE fused into the epilogue of the q|k columns
  // (col < rope_cols, heads of 128): out = rlp(x*cos) + rlp(-/+ partner*sin) with position = row % rope_S and the table
  // rope_cs [rope_S, 128] = cos(64)|sin(64) — the rounding points of rope_kernel, one pass over q,k less
  const lp_t* rope_cs; int rope_S; int rope_cols;
  int rope_R0, rope_Lc;   // grouped sequences (see attn_forward): position of row r >= rope_R0 is rope_Lc + ((r - rope_R0) & 31); 0 = off
  // shared system-prompt prefix (engine.hip::llm_forward): rows [0, rope_tail) are sequences of rope_S rows that START at position
  // rope_pos0 (their first rope_pos0 positions live once, in the rows from rope_tail on: position = row - rope_tail); 0 = off
  int rope_pos0, rope_tail;
  // gemm256 only: W8A8 mode (BASELINE config 5).  a_scale != null => A and W point at OCP fp8 e4m3 bytes (lda / K count
  // fp8 elements, K % 256 == 0), a_scale [M] and w_scale [N] are the per-row / per-output-channel dequantisation factors:
  // C = epilogue((A_q · W_q^T) * a_scale[m] * w_scale[n]).  One v_mfma_scale_f32_16x16x128_f8f6f4 (scales 1.0) replaces two
  // 16x16x32 bf16 MFMAs on the same LDS bytes, i.e. twice the K per K-tile at the same LDS/DMA traffic.
  const float* a_scale; const float* w_scale;
  // RMSNorm folded into the linears (bf16 kernels, llm_forward): `row_scale` [buffer rows] multiplies the accumulators of row r by
  // row_scale[mapped A row r] before bias / activation / RoPE — with the norm's weight folded into W's columns this IS
  // Linear(RMSNorm(x)) without the normalised copy of x.  `sumsq_out` (VSTAR_EPI_NONE, bf16 output): the epilogue also writes,
  // per output row and 64-column span, the sum of squares of the STORED bf16 values to sumsq_out[crow * sumsq_ld + col / 64] —
  // the statistics of the next RMSNorm, produced where the residual stream is written.  Both kernels add in the same fixed
  // order (4+4 columns, pairs of 8-column chunks, 16-column fragments pairwise), so the partials are bit-identical.
  const float* row_scale; float* sumsq_out; int sumsq_ld;
  // LayerNorm folded into the ViT linears (round 4): with stats_sum != 0 the epilogue ALSO writes the plain sum of every span to
  // sumsq_out[crow * sumsq_ld + stats_sum + col / 64] (stats_sum = the number of 64-column spans of a row), same fixed tree; the
  // consumer's row_scale is then 1 / sqrt(var + eps) with var = E[x^2] - E[x]^2 (ln_rstd_partials), its weights are
  // W diag(g) with every row CENTRED over k (ln_fold_weights) — sum_k x_k (w_k - mean w) = sum_k (x_k - mean x) w_k, so no shift
  // term is needed — and its bias is W b_ln + c
  int stats_sum;
  // kernel choice: 0 = the dispatcher decides, 128 / 256 = force that tile (256 fails with hipErrorInvalidValue when the shape
  // is not gemm256_eligible).  Process-wide default for 0 calls: environment VSTAR_GEMM_TILE (A/B runs).
  int tile_force;
  // gemm_skinny_ring_kernel only (round 5): a TILE-MAJOR copy of W for the decode GEMV, or null — [N / (16 NT)][K / 64][2 NT] pieces of
  // 1 KiB, each in the lane order of the LDS-DMA request that fetches it (skinny_pack_tiles): a workgroup then streams ONE sequential
  // region of HBM instead of 16 / 32 row streams 2 K bytes apart (tools/probes/stream_layout_probe.hip: gate|up 5.4 -> 7.0 TB/s)
  const lp_t* W_tiled;
  // gemm4w only, W8A8 with block-scaled activations (mx.hpp; round 6).  a_mx != null: A holds fp8 bytes whose E8M0
  // block scales (one per row and 32 k, tile-major: mx_scale_offset with m128 = M / 128) are applied inside the MFMA; w_scale as above.
  // c_mx != null (VSTAR_EPI_SILU_MUL): the epilogue writes SiLU(gate) * up as fp8 bytes to (uint8_t*)C (ldc in bytes) and the block
  // scales to c_mx (same layout, consumer K = N / 2) instead of 16-bit values — bit-identical to storing them and running
  // quantize_rows_mx over the result.  Both fail with hipErrorInvalidValue outside gemm4w's domain (gemm_mx_supported).
  const uint8_t* a_mx; uint8_t* c_mx;
  // ... and VSTAR_EPI_NONE with c_mx != null (o_proj / down_proj: residual stream out): C receives the 16-bit rows as usual AND c8
  // [M, ldc8] their block-scaled fp8 copy (scales to c_mx) — the next linear's A operand, whose RMSNorm is folded: its weight into that
  // linear's W, its 1 / rms applied as a_scale (allowed next to a_mx) from the sum-of-squares partials this epilogue writes (sumsq_out).
  uint8_t* c8; int64_t ldc8;
};
bool gemm_mx_supported(const GemmParams& p, int epilogue);
hipError_t gemm_lp(const GemmParams& p, int epilogue, bool out_f32, hipStream_t s);
// GemmParams::tile_force / gemm_last_tile() value of the 4-wave / AGPR 256 x 256 kernel (gemm4w.hip; "256, 4 waves")
constexpr int GEMM_TILE_4W = 2564;
// whether unforced calls inside gemm4w's domain take it (VSTAR_GEMM4W overrides)
#ifndef GEMM4W_DEFAULT
#define GEMM4W_DEFAULT 1
#endif
// which kernel the last gemm_lp call of THIS thread launched: 128, 256, or 0 when nothing was launched (observability for the
// op-level tests: a dispatcher change must not silently move a test onto the other kernel)
int gemm_last_tile();
// compute units of the current device, rounded down to a multiple of 8 (>= 8); cached per process
int gemm_device_cus();
// Dispatcher dry run (vstar_op_gemm_plan: host-only tests of the kernel choice).  While `on`, gemm_lp takes every decision as
// usual — with `cus` compute units instead of the device's — but launches nothing; gemm_last_tile() / gemm_last_mode() then
// tell what it would have launched.  Thread-local, like gemm_last_tile().
void gemm_set_plan(bool on, int cus);
bool gemm_plan_only();
// 128-family variant of the last gemm_lp call of this thread: 2 = double buffer, 5 / 6 / 7 = loader-wave ring on the 128 x 128 /
// 128 x 64 / 128 x 256 tile; 0 when the last call did not reach that family
int gemm_last_mode();
bool gemm256_eligible(const GemmParams& p);   // true: gemm_lp runs the 256^2 kernel (the only one that honours rope_cs)

// decode-sized GEMM (decode.hip): M <= 64, identity row maps; same operands/epilogues as gemm_lp
bool gemm_skinny_eligible(const GemmParams& p);
hipError_t gemm_skinny_lp(const GemmParams& p, int epilogue, bool out_f32, hipStream_t s);
// Wt <- tile-major image of the packed row-major W [n_rows][K] for gemm_skinny_ring_kernel (nt = 2 for SiLU(gate)*up weights whose
// rows interleave gate / up in blocks of 16, else 1); n_rows % (16 nt) == 0, K % 64 == 0; Wt holds n_rows * K elements
hipError_t skinny_pack_tiles(const lp_t* W, lp_t* Wt, int n_rows, int K, int nt, hipStream_t s);

// ---- KV-cached language model + Perceiver resampler (decode.hip) ----
// x[r,:] = src[r] >= 0 ? table[src[r]] : (src[r] == INT32_MIN ? 0 : feats[-(src[r]+1)])
hipError_t embed_rows(const int32_t* src, const lp_t* table, int vocab, const lp_t* feats, int64_t n_feat_rows, lp_t* x, int R,
                      int C, hipStream_t s);
// in-place RoPE on q,k of qkv [R, 3*H*128] at row_pos[r] (rows with row_pos < 0 are skipped) and K/V rows appended to the
// cache of slot row_slot[r]: kc/vc = this layer's [slot][H][ctx][128]
hipError_t rope_kv_append(lp_t* qkv, const lp_t* cos_sin, const int32_t* row_pos, const int32_t* row_slot, lp_t* kc, lp_t* vc,
                          int64_t slot_stride, int ctx, int R, int H, hipStream_t s);
// causal attention of R new rows against the cache: keys [0, seq_past) come from seq_prefix's slot, the rest from seq_kv's.
// fused_cos_sin != null (only when every sequence has exactly ONE new row): the kernel also does rope_kv_append's work for
// its row (RoPE on q,k, K/V appended to the cache) — qkv then holds the raw projection.
hipError_t cached_attention(const lp_t* qkv, lp_t* kc, lp_t* vc, const int32_t* row_seq, const int32_t* row_pos,
                            const int32_t* seq_kv, const int32_t* seq_prefix, const int32_t* seq_past, const lp_t* fused_cos_sin,
                            lp_t* out, int R, int H, int ctx, int64_t slot_stride, int max_keys, hipStream_t s,
                            void* split_ws = nullptr, int split_max_rows = 0);
// workspace of the split-KV decode path (scores, partial statistics / outputs, tickets) for up to max_rows new rows per step;
// must be zero-filled once
size_t cached_attention_split_ws_bytes(int max_rows, int H, int ctx);
// q [n*L, H*DH], kv [n*NK, 2*H*DH] (k | v) -> out [n*L, H*DH]
hipError_t perceiver_attention(const lp_t* q, const lp_t* kv, lp_t* out, int n, int L, int NK, int H, int DH, hipStream_t s);
hipError_t argmax_rows_lp(const lp_t* x, int rows, int cols, int64_t ld, int32_t* out, hipStream_t s);

// ---- W8A8 (quant.hip): per-row symmetric fp8 e4m3 quantisation, scale = absmax / 448 ----
hipError_t quantize_rows_fp8(const lp_t* x, int64_t ldx, uint8_t* q, int64_t ldq, float* scale, int rows, int cols,
                             hipStream_t s);
// LlamaRMSNorm whose 16-bit output row is quantised on the way out (cols <= 4096)
hipError_t rmsnorm_quant_fp8(const lp_t* x, const lp_t* gamma, uint8_t* q, float* scale, int rows, int cols, float eps,
                             hipStream_t s);

// block-scaled twin (mx.hpp): x [rows, cols] 16-bit -> fp8 bytes q [rows, ldq] + E8M0 scales (tile-major, rows % 128 == 0, cols % 128 == 0)
hipError_t quantize_rows_mx(const lp_t* x, int64_t ldx, uint8_t* q, int64_t ldq, uint8_t* scales, int rows, int cols, hipStream_t s);

// ---- norms (norm.hip) ----
// y[r] = LN(x[row_index ? row_index[r] : r]) ; act: 0 none, 1 exact GELU after the affine (LayerNorm2d+GELU)
hipError_t layernorm_lp(const lp_t* x, const lp_t* gamma, const lp_t* beta, lp_t* y, int rows, int cols,
                          float eps, const int32_t* row_index, int act, hipStream_t s);
// r[row] = rsqrt(mean(x[row]^2) + eps) (LlamaRMSNorm's fp32 statistics) — from x itself, or from the 64-column partial sums a GEMM
// epilogue wrote (GemmParams::sumsq_out); the two agree bit for bit (same summation tree).  cols % 64 == 0.
hipError_t rms_rstd_rows(const lp_t* x, int rows, int cols, float eps, float* r, hipStream_t s);
hipError_t rms_rstd_partials(const float* partials, int ld, int rows, int cols, float eps, float* r, hipStream_t s);
// LayerNorm twins: r[row] = 1 / sqrt(E[x^2] - E[x]^2 + eps) from the rows themselves / from the epilogue's partials (sumsq spans
// at [0, cols/64), sum spans at [cols/64, 2 cols/64) of each partial row); same summation trees, bit-identical to each other
hipError_t ln_rstd_rows(const lp_t* x, int rows, int cols, float eps, float* r, hipStream_t s);
hipError_t ln_rstd_partials(const float* partials, int ld, int rows, int cols, float eps, float* r, hipStream_t s);
// Folds LayerNorm(g, b_ln) into the packed Linear (W [n_rows, K] K-contiguous, bias [n_rows] or null) that consumes it, in place:
// bias[n] = sum_k W[n,k] b_ln[k] + bias[n] (fp32, from the unscaled W), then W[n,:] = W[n,:] * g - mean_k(W[n,:] * g)
hipError_t ln_fold_weights(lp_t* W, lp_t* bias, const lp_t* g, const lp_t* b_ln, int n_rows, int K, hipStream_t s);
// W[n, k] = round(W[n, k] * w[k]) for n < rows: folds a norm weight into the columns of a packed Linear; fill_lp: v[i] = value
hipError_t scale_cols_lp(lp_t* W, const lp_t* w, int64_t rows, int K, hipStream_t s);
hipError_t fill_lp(lp_t* v, int64_t n, float value, hipStream_t s);
hipError_t rmsnorm_lp(const lp_t* x, const lp_t* gamma, lp_t* y, int rows, int cols, float eps,
                        const int32_t* row_index, hipStream_t s);

// ---- attention (attention.hip) ----
// qkv: [B*S, 3*H*D] (q | k | v).  attn_prepare: in-place rotate-half RoPE on q,k (no-op when cos_sin == null)
// Grouped sequences (grp_R0 > 0; causal D = 128 only): rows [0, grp_Lc) of every sequence are a shared prefix, rows [grp_R0, S) are
// independent 32-row suffix blocks that attend to the prefix and causally to themselves; RoPE positions of a suffix block restart
// at grp_Lc.  grp_R0 % 128 == 0, (S - grp_R0) % 32 == 0.
hipError_t attn_prepare(lp_t* qkv, const lp_t* cos_sin /*[S, D] = cos(D/2)|sin(D/2), or null*/, int B, int S, int H, int D,
                        hipStream_t s, int grp_R0 = 0, int grp_Lc = 0);
hipError_t attn_forward(const lp_t* qkv, lp_t* out, int B, int S, int H, int D, int causal, float scale, hipStream_t s,
                        int grp_R0 = 0, int grp_Lc = 0);
// causal D = 128 with the output written block-scaled (mx.hpp): fp8 bytes out8 [B*S, H*128] + E8M0 scales (m128 = B*S / 128; B*S % 128
// == 0) — bit-identical to attn_forward followed by quantize_rows_mx
hipError_t attn_forward_mx(const lp_t* qkv, uint8_t* out8, uint8_t* scales, int B, int S, int H, float scale, hipStream_t s,
                           int grp_R0 = 0, int grp_Lc = 0);
// generic small attention for the SAM head: q[B,Nq,H*D] k[B,Nk,H*D] v[B,Nk,H*D] -> out[B,Nq,H*D]; D <= 32, fp32 math
hipError_t small_attention(const lp_t* q, const lp_t* k, const lp_t* v, lp_t* out, int B, int Nq, int Nk, int H,
                           int D, hipStream_t s);

// ---- elementwise / layout (elementwise.hip) ----
// pix [B,3,I,I] -> A [B*P, Kpad], k = c*ps*ps + ky*ps + kx, zero padded to Kpad
hipError_t im2col_patch(const lp_t* pix, lp_t* A, int B, int I, int ps, int Kpad, hipStream_t s);
// tokens[b,0]=cls+pos[0]; tokens[b,1+p]=patch[b,p]+pos[1+p]   (bf16 add)
hipError_t vit_assemble_tokens(const lp_t* patch, const lp_t* cls, const lp_t* pos, lp_t* tokens, int B, int P,
                               int C, hipStream_t s);
// LLaMA input embeddings for the text positions of the spliced sequence (image rows are written by the projector GEMM)
hipError_t llm_embed_text(const int32_t* ids, int L, int img_col, int P, const lp_t* table, int vocab, lp_t* x, int B,
                          int C, hipStream_t s);
// out[r, :] = a[r, :] + b[(r % b_rows), :]   (bf16 add; b broadcast over groups of b_rows)
hipError_t add_bcast(const lp_t* a, const lp_t* b, lp_t* out, int64_t rows, int cols, int64_t b_rows, hipStream_t s);
// out[(n * rows_per + p), :] = a[((n / rep) * rows_per + p), :] + b[0, :]  for n < n_out : every block of rows_per rows repeated rep times
hipError_t add_bcast_repeat(const lp_t* a, const lp_t* b, lp_t* out, int n_out, int rep, int rows_per, int cols, hipStream_t s);
// dst[(r * rep_stride + i) * ld + 0..cols) = src[i * ld + 0..cols) for r < nrep, i < nrows (one block of rows copied to the head of
// every sequence); cols % 8 == 0
hipError_t bcast_rows(const lp_t* src, lp_t* dst, int nrep, int64_t rep_stride, int nrows, int cols, int64_t ld, hipStream_t s);
// OWL-ViT: y[b,p,:] = x[b,1+p,:] * x[b,0,:]  (x = post_layernorm output, [B,N,C]) -> [B,N-1,C]
hipError_t owl_cls_mul(const lp_t* x, lp_t* y, int B, int N, int C, hipStream_t s);
// gather rows: y[r,:] = x[idx[r],:]
hipError_t gather_rows(const lp_t* x, const int32_t* idx, lp_t* y, int rows, int cols, hipStream_t s);
// argmax over fp32 logits rows
hipError_t argmax_rows(const float* x, int rows, int cols, int ld, int32_t* out, int out_stride, hipStream_t s);

// ---- heads (heads.hip) ----
// class head: emb [R, ldc] fp32 = dense0(512) | shift | scale ; query [B, Q] bf16; rows_per_crop = 2304
// img_div: record b reads the image rows of crop b / img_div (several queries per crop: grouped scoring); 1 = one query per crop
hipError_t owl_class_logits(const float* emb, int ld, int Q, const lp_t* query, float* out, int out_stride_crop,
                            int B, int rows_per_crop, hipStream_t s, int img_div = 1);
// box head final: raw [R, 4] fp32 (dense2 out incl. bias) + grid bias -> sigmoid -> out[b*stride + p*4 ..]
hipError_t owl_box_finish(const float* raw, int ld, float* out, int out_stride_crop, int B, int grid, hipStream_t s, int img_div = 1);
// SAM upscaling: bilinear x2 (align_corners=False, fp32 -> bf16) fused with 3x3 im2col (zero pad):
// src [B, h, w, C] channels-last -> A [B*(2h)*(2w), 9*C], k = (ky*3+kx)*C + c
hipError_t upsample2x_im2col3x3(const lp_t* src, lp_t* A, int B, int h, int w, int C, hipStream_t s);
// masks[b, pix] = sum_c hyper[b,c] * up[b,pix,c]  (bf16 in, fp32 accumulate, bf16-rounded like the reference matmul) -> fp32
hipError_t hyper_mask(const lp_t* hyper, const lp_t* up, float* out, int out_stride_crop, int B, int npix, int C,
                      hipStream_t s);
// bilinear resize (align_corners=False) of fp32 [hin,win] -> [hout,wout], then clamp(min=0) unless clamp_min0 == 0
hipError_t resize_bilinear_clamp(const float* in, int hin, int win, float* out, int hout, int wout, hipStream_t s, int clamp_min0 = 1);

// min / max / total / rectangle sums of clamp(bilinear(lowres -> hout x wout), 0) for n maps in ONE launch (device arrays: lowres
// [n][hin*win], hw [n][2] = (hout, wout), rects [n][32], n_rects [n], out [n][3 + 8] doubles, mm_scratch [n][2])
hipError_t heat_stats_batch(const float* lowres, int hin, int win, const int* hw, const int* rects, const int* n_rects, int n,
                            int64_t max_pixels, double* out, unsigned* mm_scratch, hipStream_t s);

// ---- GPU-side crop preprocessing (preprocess.hip) ----
struct PreJob {            // one (crop, target) pair; jobs are stored as [crop][0 = CLIP, 1 = OWL-ViT]
  int x0, y0, cw, ch;      // crop box inside the resident image
  int in_w, in_h;          // resampled extent (CLIP: the padded square side; OWL: cw, ch)
  int out;                 // output side (I or 768)
  int hb_off, hc_off, hks; // horizontal bounds / coefficients offsets (int32 units) and kernel size
  int vb_off, vc_off, vks;
  int img_w;               // row pitch (pixels) of the image this crop is cut from
  int64_t temp_off;        // byte offset of this job's uint8 intermediate [in_h][out][3]
  int64_t out_off;         // element offset into the bf16 pixel buffer [3][out][out]
  const uint8_t* img;      // the resident image (slot) this crop is cut from
};
void pil_bicubic_coeffs(int in_size, int out_size, std::vector<int32_t>* bounds, std::vector<int32_t>* coeffs, int* ksize);
void clip_norm_lut(lp_t* lut);
hipError_t preprocess_launch(const PreJob* jobs, const int32_t* tables, uint8_t* temp,
                             const lp_t* lut, lp_t* out, int which, int B, int out_size, int max_in_h, hipStream_t s);

}  // namespace VS_NS
using namespace VS_NS;
