/*
 * vstar_hip.h — C-ABI of libvstar_hip.so, the MI355X (gfx950) engine for the V* guided-visual-search
 * hot path: per-crop VSM scoring = CLIP-ViT-L/14 -> mm_projector -> LLaMA-7B prefill -> [LOC] hidden state
 * -> text_hidden_fcs_{det,seg} -> OWL-ViT tower + class/box heads -> SAM-style mask head.
 *
 * The reference (penghao-wu/vstar) is 100 % Python and has no FFI of its own; the Python call sites
 * this library replaces are cited per entry point (paths relative to the reference repo root).
 *
 * Conventions
 *   - return 0 = OK, negative = error; vstar_last_error() gives the message for the handle
 *     (or for the calling thread when the handle is NULL / creation failed).
 *   - The caller owns every host buffer; the engine owns every device buffer.  Pointers named dev_* are
 *     device (HBM) pointers supplied by the caller (e.g. a torch tensor's data_ptr()); everything else is host.
 *   - One handle per device/process, not thread-safe.  Work is stream-ordered on the handle's HIP stream;
 *     entry points synchronise that stream before returning unless they say otherwise.
 *   - bf16 tensors cross the ABI as uint16_t (raw bf16 bits).  No torch / Python types in any signature.
 */
#ifndef VSTAR_HIP_H
#define VSTAR_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VSTAR_ABI_VERSION 1

/* dtype codes for vstar_load_tensor */
enum { VSTAR_F32 = 0, VSTAR_F16 = 1, VSTAR_BF16 = 2 };

/* error codes */
enum {
  VSTAR_OK = 0,
  VSTAR_ERR_INVALID = -1,   /* bad argument / shape */
  VSTAR_ERR_STATE = -2,     /* call order (e.g. score before finalize) */
  VSTAR_ERR_MISSING = -3,   /* a required checkpoint tensor was never loaded */
  VSTAR_ERR_HIP = -4,       /* HIP runtime error */
  VSTAR_ERR_NOMEM = -5
};

/*
 * Model geometry.  Mirrors what the reference reads from the HF configs:
 *   LlavaConfig / LlamaConfig            VisualSearch/model/llava/model/language_model/llava_llama.py:31-52
 *   CLIPVisionConfig (vision tower)      VisualSearch/model/llava/model/multimodal_encoder/clip_encoder.py:17-29
 *   OwlViTConfig.vision_config           VisualSearch/model/owlvit/owlvit.py:21-31
 *   SAM head hyper-parameters            VisualSearch/model/VSM.py:91-113 (fixed: 256-d, 48x48 grid, 8 heads, mlp 2048)
 * Head dims are fixed by the kernels: CLIP/OWL-ViT 64, LLaMA 128.
 */
typedef struct vstar_config {
  int32_t abi_version;        /* VSTAR_ABI_VERSION */
  /* CLIP vision tower */
  int32_t clip_image_size;    /* 224 (reference) or 336 (benchmark geometry) */
  int32_t clip_patch;         /* 14 */
  int32_t clip_hidden;        /* 1024 */
  int32_t clip_heads;         /* 16 (hidden / 64) */
  int32_t clip_mlp;           /* 4096 */
  int32_t clip_layers;        /* 24 in the checkpoint; blocks 0..clip_layers+select_layer are executed */
  int32_t clip_select_layer;  /* -2: hidden_states[-2] => clip_layers-1 blocks */
  /* LLaMA decoder */
  int32_t llm_hidden;         /* 4096 */
  int32_t llm_heads;          /* 32 (hidden / 128) */
  int32_t llm_mlp;            /* 11008 */
  int32_t llm_layers;         /* 32 */
  int32_t llm_vocab;          /* 32004 */
  float   llm_rms_eps;        /* 1e-6 (vicuna-7b-v1.3) */
  float   llm_rope_theta;     /* 10000 */
  /* OWL-ViT vision tower + heads */
  int32_t owl_image_size;     /* 768 */
  int32_t owl_patch;          /* 16 */
  int32_t owl_hidden;         /* 768 */
  int32_t owl_heads;          /* 12 */
  int32_t owl_mlp;            /* 3072 */
  int32_t owl_layers;         /* 12 */
  int32_t owl_query_dim;      /* 512 = text_hidden_fcs_det out_dim */
  /* limits used to size the workspace */
  int32_t max_batch;          /* crops per vstar_vsm_score_batch call */
  int32_t max_text_len;       /* L_max: input_ids length incl. the single -200 */
  int32_t llm_w8a8;           /* 1: LLaMA linears run W8A8 on the fp8 MFMA (BASELINE config 5) whenever a call has >= 1024 rows;
                                 per-output-channel weight scales, per-token activation scales, OCP e4m3.  0 (default): bf16 */
  int32_t reserved[7];
} vstar_config;

typedef struct vstar_engine vstar_handle;

/* Per-crop result record.  Fixed size so that records can be all-gathered across ranks unchanged
 * (SURVEY.md §8e).  n_patches = (owl_image_size/owl_patch)^2 = 2304, low-res mask = 192x192. */
#define VSTAR_N_BOXES 2304
#define VSTAR_MASK_RES 192
#define VSTAR_MAX_VERIFY 8
typedef struct vstar_result {
  float   pred_logits[VSTAR_N_BOXES];            /* raw class logits (pre-sigmoid)  VSM.py:544-552 */
  float   pred_boxes[VSTAR_N_BOXES * 4];         /* cxcywh in [0,1]                 owlvit.py:79-100 */
  float   lowres_mask[VSTAR_MASK_RES * VSTAR_MASK_RES]; /* mask 0 logits            mask_decoder.py:138-186 */
  int32_t tf_argmax[VSTAR_MAX_VERIFY];           /* argmax(lm_head(h_t)) at the verify positions */
} vstar_result;

/* Replaces VSMForCausalLM.from_pretrained(...) construction  (visual_search.py:143-172). */
int vstar_create(const vstar_config* cfg, int device, vstar_handle** out);
void vstar_destroy(vstar_handle* h);
const char* vstar_last_error(const vstar_handle* h);
/* 16 hex digits: hash of the kernel sources (the .hip / .hpp files of vstar_amd/csrc and build.sh) this library was BUILT from.  No reference
 * counterpart; evidence files under profiles/ are stamped with it (vstar_amd/provenance.py). */
const char* vstar_build_source_hash(void);

/* Weight hand-over, one checkpoint tensor at a time, keyed by its HF state-dict name
 * (VSM keys as listed in SURVEY.md §5; the CLIP tower's keys carry the prefix "clip.").
 * host_ptr is copied; dtype is one of VSTAR_F32/F16/BF16.  Replaces the torch state-dict load
 * inside from_pretrained (visual_search.py:157-161). */
int vstar_load_tensor(vstar_handle* h, const char* hf_key, const void* host_ptr, int dtype,
                      int ndim, const int64_t* shape);
/* Packs weights into the kernels' layouts (fused QKV, interleaved gate/up, padded N/K), uploads,
 * and frees host staging copies.  Fails with VSTAR_ERR_MISSING naming the first absent key. */
int vstar_finalize_weights(vstar_handle* h);

/*
 * The hot path: score B crops in one pass.  Replaces VSMForCausalLM.inference / model_forward(inference=True)
 * (VisualSearch/model/VSM.py:438-553, 201-364) called from VSM.inference (visual_search.py:198-207).
 *   clip_pix  [B,3,I,I]      bf16, CLIP-normalised      (visual_search.py:186-189)
 *   owl_pix   [B,3,768,768]  bf16, OWL-ViT-normalised   (visual_search.py:190-194)
 *   ids       [B,L] int32, exactly one -200 (IMAGE_TOKEN_INDEX) per row at the same column
 *   loc_pos   [B]   index into the SPLICED sequence (length L-1+P) of the hidden state that predicts [LOC]
 *                   (= index([LOC]) - 1 + (P-1), VSM.py:465-473)
 *   verify_pos [B*n_verify] spliced-sequence positions whose lm_head argmax is returned (template check,
 *                   SURVEY.md §7 "hard parts"); n_verify may be 0.
 *   flags     bit0: skip the OWL-ViT tower + heads (core path only; result boxes/logits/mask untouched)
 * pixel/ids pointers are HOST pointers unless flags bit1 (VSTAR_F_DEVICE_INPUTS) is set.
 * out: B records in host memory (or device memory when VSTAR_F_DEVICE_OUTPUT is set).
 */
#define VSTAR_F_SKIP_OWL       1u
#define VSTAR_F_DEVICE_INPUTS  2u
#define VSTAR_F_DEVICE_OUTPUT  4u
#define VSTAR_F_NO_SYNC        8u   /* do not synchronise the stream before returning (bench inner loop) */
#define VSTAR_F_INTERNAL_PIXELS 16u /* pixels were produced on the device by vstar_preprocess_crops (clip_pix/owl_pix ignored) */
/* Shared system prompt (every crop of a search carries the same text before <image>: visual_search.py:176-183 builds the prompt
 * from conv_templates["llava_v1"] + the question).  When all B rows of `ids` agree on the >= 16 tokens before the image token, those
 * positions go through LLaMA ONCE (one extra prefix "sequence") instead of B times: under the causal mask their states do not depend
 * on what follows.  The B sequences' linears then run on their remaining rows only (GEMM row maps); the prefix's q|k|v rows are
 * copied to the head of every sequence before the attention.  A crop's record does not depend on B or on the other crops (the
 * prefix is always computed as its own sequence, also for B = 1); against a call without the flag it is a second evaluation of the
 * same numbers (identical except where the attention's softmax re-centring falls differently).  Ignored in W8A8 mode.
 * Measured at 32 crops / 7B on 256 CUs: no faster (5 % fewer GEMM rows = 76 instead of 80 row tiles = the same number of whole
 * rounds of 256x256 tiles); opt-in for batch sizes where the saved rows cross a round boundary. */
#define VSTAR_F_SHARE_PREFIX   32u
int vstar_vsm_score_batch(vstar_handle* h, int B, const uint16_t* clip_pix, const uint16_t* owl_pix,
                          const int32_t* ids, int L, const int32_t* loc_pos,
                          const int32_t* verify_pos, int n_verify, unsigned flags, vstar_result* out);

/*
 * Grouped scoring: G crops x T prompts per crop whose tokenised prompts share their first Lp ids (system prompt, the -200 image
 * placeholder and the common start of the question — for the search loop everything up to the target's name), e.g. several
 * search targets evaluated on the same crops (vstar_bench_eval.py:205-209 loops visual_search per missing object).  Under the
 * causal mask the states of the shared positions do not depend on what follows them, so they are computed ONCE per crop
 * (CLIP tower, projector, and Lp - 1 + P of the LLaMA rows), each prompt only adds its suffix rows (<= 32), and the OWL-ViT
 * tower / box head run once per crop.  Record n = g * T + t equals what vstar_vsm_score_batch returns for (crop g, prompt t)
 * up to bf16 rounding (the attention visits the keys in a different tiling).
 *   prefix_ids [Lp]            shared ids, exactly one -200
 *   suffix_ids [G*T*Ls]        per-record continuation, right-padded to Ls <= 32 with any valid id
 *   loc_in_suffix [G*T]        index INSIDE the suffix of the token that precedes [LOC]
 *   verify_in_suffix [G*T*nv]  indices inside the suffix whose next-token arg-max is returned in tf_argmax
 * Requires G * T <= max_batch and G * (round_up(Lp - 1 + P, 128) + 32 T) <= max_batch * (max_text_len - 1 + P).
 * Pixels are those of the G crops ([G,3,I,I] / [G,3,768,768]); flags as for vstar_vsm_score_batch (no VSTAR_F_SKIP_OWL).
 */
int vstar_vsm_score_grouped(vstar_handle* h, int G, int T, const uint16_t* clip_pix, const uint16_t* owl_pix,
                            const int32_t* prefix_ids, int Lp, const int32_t* suffix_ids, int Ls, const int32_t* loc_in_suffix,
                            const int32_t* verify_in_suffix, int n_verify, unsigned flags, vstar_result* out);

/* GPU-side preprocessing.  vstar_image_set uploads the full RGB uint8 image [height,width,3] once;
 * vstar_preprocess_crops turns B crop boxes (x0,y0,x1,y1 exactly as passed to PIL `image.crop`, visual_search.py:394)
 * into the engine's CLIP and OWL-ViT input tensors: expand2square (top-left, CLIP-mean colour) + resize IxI, and resize to
 * 768x768, both bit-identical to PIL.Image.resize(BICUBIC) on uint8, then the HF rescale/normalise -> bf16
 * (replaces visual_search.py:186-194).  Follow with vstar_vsm_score_batch(..., VSTAR_F_INTERNAL_PIXELS). */
int vstar_image_set(vstar_handle* h, const uint8_t* rgb, int height, int width);
int vstar_preprocess_crops(vstar_handle* h, int B, const int32_t* boxes_xyxy);

/* Image SLOTS (round 3): several full images resident in HBM at once, so that one engine batch can hold crops of DIFFERENT
 * images — the cross-image lock-step search (vstar_amd/search.py::visual_search_stream) keeps a window of concurrent
 * (image, target) searches and fills every batch with the crops they need next, instead of the reference's one-sample-at-a-time
 * loop (visual_search.py:536-560, vstar_bench_eval.py:190-262).  vstar_image_set == slot 0; vstar_preprocess_crops == every box
 * from slot 0.  slot in [0, VSTAR_MAX_IMAGE_SLOTS); `slots` [B] names the image of each box (null: all slot 0).  A 4K image is
 * 25 MB: 64 slots are 1.6 GB of the 288 GB. */
#define VSTAR_MAX_IMAGE_SLOTS 64
int vstar_image_set_slot(vstar_handle* h, int slot, const uint8_t* rgb, int height, int width);
int vstar_preprocess_crops_slots(vstar_handle* h, int B, const int32_t* boxes_xyxy, const int32_t* slots);
/* Round 4: the same upload WITHOUT stalling the scoring stream — `rgb` is copied into a pinned staging buffer before the call
 * returns (the caller may free it), the DMA runs on a copy stream of its own, and the next vstar_preprocess_crops* that reads the
 * slot waits for it on the device.  Meant to be called from a second host thread while another thread is inside a scoring call:
 * the stream search uploads the images of the samples that enter the window NEXT while the current step is on the GPU (in the
 * reference the equivalent host work — Image.open + processors, visual_search.py:541-550 — sits between two forward passes).
 * Contract: no crop of the slot's previous image may still be waiting to be launched (crops already launched are ordered before
 * the copy); ONE asynchronous upload at a time per handle (the staging ring is not re-entrant: `VSM.set_image_async` serialises
 * its callers), concurrently with at most one scoring / preprocessing call on another thread. */
int vstar_image_set_slot_async(vstar_handle* h, int slot, const uint8_t* rgb, int height, int width);

/* Greedy free-text decode of ONE crop with a KV cache — VSMForCausalLM.inference for mode='vqa' (VSM.py:438-462 ->
 * generate(max_new_tokens, greedy); called from VSM.inference at visual_search.py:198-219 for the contextual-cue branch,
 * :427-443).  The reference re-runs the whole prefix for every new token (use_cache=False); here the prompt is prefilled
 * once (CLIP tower + projector + LLaMA prefill, K/V kept in HBM) and each new token is one weight-streaming decode step,
 * which yields the same arg-max tokens.  clip_pix: [1,3,I,I] bf16 (host, or device with VSTAR_F_DEVICE_INPUTS); ids[L]:
 * the prompt with one -200; generation stops after max_new_tokens or at eos_id (which is stored as the last id).
 * out_ids must hold max_new_tokens entries; *n_out receives the count. */
int vstar_vsm_generate(vstar_handle* h, const uint16_t* clip_pix_bf16, const int32_t* ids, int L, int max_new_tokens,
                       int eos_id, unsigned flags, int32_t* out_ids, int32_t* n_out);

/* Bilinear (align_corners=False) upsample of a 192x192 low-res mask to h_out x w_out fp32, then clamp(min=0).
 * Replaces F.interpolate(...) + torch.clamp (VSM.py:534-537, visual_search.py:223-224). Host in, host out. */
int vstar_upsample_mask(vstar_handle* h, const float* lowres, int h_out, int w_out, float* out);
/* Same with the clamp optional (clamp_min0 = 0: the bare F.interpolate of VSM.py:534-536, which is what the model-level
 * VSMForCausalLM.inference returns — the clamp belongs to its caller, visual_search.py:211,223). */
int vstar_upsample_mask_ex(vstar_handle* h, const float* lowres, int h_out, int w_out, int clamp_min0, float* out);

/* Decision statistics of a heat map without materialising it: with H = clamp(bilinear(lowres 192x192 ->
 * h_out x w_out, align_corners=False), 0) writes out[0] = min H, out[1] = max H, out[2] = sum H, out[3+k] = sum of H over
 * rectangle k (x, y, w, h in output pixels; n_rects <= 8).  fp64 accumulation.  These are the quantities
 * visual_search.py:420-426 (max, min-max normalisation) and :255-266 (get_subpatch_scores) consume. Host in, host out. */
int vstar_heatmap_stats(vstar_handle* h, const float* lowres, int h_out, int w_out, int n_rects, const int32_t* rects_xywh,
                        double* out);
/* n heat maps in one call (one upload, n kernel passes, one download, one synchronisation): item i is lowres[i] (192x192),
 * out_hw[2i], out_hw[2i+1] = (h_out, w_out), n_rects[i] <= 8 rectangles at rects_xywh[32 i ...]; writes out[11 i ...] laid out like
 * vstar_heatmap_stats' `out` (unused rectangle slots are 0).  The scheduler asks for a node's own statistics and its
 * ancestors' (visual_search.py:445-462 accumulates over all ancestors) — and, in a lock-step multi-target search, every
 * target's — together. */
int vstar_heatmap_stats_batch(vstar_handle* h, int n, const float* lowres, const int32_t* out_hw, const int32_t* n_rects,
                              const int32_t* rects_xywh, double* out);

/* Debug/parity taps: copy an internal activation of the LAST score_batch call to host as fp32.
 * name in {"clip_features","projector","llm_hidden_loc","embed_det","embed_seg","owl_feats"}.
 * Returns the number of floats written (<= cap) or a negative error. */
int64_t vstar_debug_read(vstar_handle* h, const char* name, float* out, int64_t cap);

/* Stream handle (hipStream_t) the engine launches on, for HIP-event timing by the caller. */
void* vstar_stream(vstar_handle* h);

/* The data-parallel exchange of the search loop: one process per GPU, each engine step's crops dealt round-robin
 * over the ranks, then ONE all-gather of the fixed-size records so that every rank takes the same next-step decision
 * (replaces nothing in the reference, which is single-GPU: visual_search.py:520-566; it is what BASELINE configs 3/4 add).
 * RCCL (ncclAllGather over xGMI) on the ENGINE'S stream, bound at run time (dlopen; no link-time dependency).
 *   vstar_comm_unique_id : rank 0 obtains the 128-byte id and hands it to the other ranks by the host's own means
 *                          (torch.distributed store, MPI, a file: vstar_amd/dist.py::engine_comm_init shows the torch way)
 *   vstar_comm_init      : collective over all `world` ranks (ncclCommInitRank on the engine's device)
 *   vstar_allgather_results : local_dev [n_local] -> gathered_dev [world * n_local] (rank-major; both DEVICE pointers, e.g. the
 *                          VSTAR_F_DEVICE_OUTPUT buffer of vstar_vsm_score_batch); every rank passes the same n_local (pad the
 *                          last shard); flags: VSTAR_F_NO_SYNC leaves the gather in flight on the engine's stream
 *   vstar_comm_destroy   : also called by vstar_destroy */
#define VSTAR_COMM_ID_BYTES 128
int vstar_comm_unique_id(uint8_t* id_out /* [VSTAR_COMM_ID_BYTES] */);
int vstar_comm_init(vstar_handle* h, const uint8_t* id, int world, int rank);
int vstar_allgather_results(vstar_handle* h, const vstar_result* local_dev, int n_local, vstar_result* gathered_dev, unsigned flags);
int vstar_comm_destroy(vstar_handle* h);

/* Last-call kernel timing: the engine brackets the dominant GEMM family with HIP events when enabled.
 * Returns accumulated GEMM kernel milliseconds and launch count since the last reset. */
int vstar_profile_enable(vstar_handle* h, int on);
int vstar_profile_read(vstar_handle* h, double* gemm_ms, int64_t* gemm_launches, double* gemm_flops);
/* The W8A8 subset of the same profile: the launches that ran on the fp8 MFMA (the LLaMA linears of a handle created with
 * llm_w8a8 = 1), whose roofline is the ~5 PFLOP/s fp8 peak; vstar_profile_read's totals include them. */
int vstar_profile_read_fp8(vstar_handle* h, double* gemm_ms, int64_t* gemm_launches, double* gemm_flops);

/* ------------------------------------------------------------------------------------------------
 * Operator-level entry points (device pointers).  These are the kernels the engine is built from,
 * exported so that parity tests can drive each one against the oracle through the same C-ABI.
 * All tensors are dense row-major; "ld" = leading dimension in elements.  stream may be NULL.
 * ---------------------------------------------------------------------------------------------- */
enum { VSTAR_EPI_NONE = 0, VSTAR_EPI_QUICK_GELU = 1, VSTAR_EPI_GELU = 2, VSTAR_EPI_RELU = 3, VSTAR_EPI_SILU_MUL = 4,
       VSTAR_EPI_NOSYNC = 0x100, /* OR-ed into `epilogue`: launch only, do not synchronise (micro-benchmarks) */
       VSTAR_EPI_TILE128 = 0x200, /* OR-ed into `epilogue`: force the 128x128 kernel for this call */
       VSTAR_EPI_TILE256 = 0x400, /* OR-ed into `epilogue`: force the 8-wave 256x256 kernel; VSTAR_ERR_INVALID when the shape is outside
                                     its domain (M >= 1024, N >= 256, K % 128 == 0) — never silently re-routed */
       VSTAR_EPI_TILE4W = 0x800   /* OR-ed into `epilogue`: force the 4-wave / AGPR 256x256 kernel (gemm4w.hip); error outside its
                                     domain (M % 256 == 0, N % 256 == 0, K % 128 == 0, 16-byte aligned operands, bf16/fp16 output) */ };

/* C[M,N] = epi(A[M,K] @ W[N,K]^T + bias) (+ residual).  bf16 in, fp32 accumulate (MFMA), bf16 or fp32 out.
 * Replaces every nn.Linear / conv-as-GEMM on the path (SURVEY.md §8d GEMM shape list).
 * W must have ceil(N/256)*256 rows allocated (rows >= N are never stored) and K % 64 == 0.
 * For VSTAR_EPI_SILU_MUL, W holds gate/up rows interleaved in blocks of 16 and the output has N/2 columns. */
int vstar_op_gemm(void* stream, const uint16_t* dev_A, int64_t lda, const uint16_t* dev_W, const uint16_t* dev_bias,
                  const uint16_t* dev_residual, int64_t ldr, void* dev_C, int64_t ldc, int out_f32,
                  int M, int N, int K, int epilogue);
/* Op-level door for the RMSNorm-folded linears of the LLaMA pass (bf16 output): C = epilogue((A . W^T) * row_scale[m] + bias)
 * (+ residual); with sumsq_out (VSTAR_EPI_NONE, N % 64 == 0) the epilogue also writes sumsq_out[m * sumsq_ld + n / 64] = the sum
 * of squares of the 64 stored values C[m, n .. n+63] (fixed summation order, identical in both GEMM kernels).  Either may be null.
 * vstar_op_rms_rstd: r[m] = rsqrt(mean(x[m]^2) + eps) from x [rows, cols] (x != null) or from such partial sums
 * (x == null: partials [rows, ld], cols / 64 of them per row) — bit-identical either way. */
int vstar_op_gemm_norm(void* stream, const uint16_t* dev_A, int64_t lda, const uint16_t* dev_W, const uint16_t* dev_bias,
                       const uint16_t* dev_residual, int64_t ldr, void* dev_C, int64_t ldc, int M, int N, int K, int epilogue,
                       const float* dev_row_scale, float* dev_sumsq_out, int sumsq_ld);
int vstar_op_rms_rstd(void* stream, const uint16_t* dev_x, const float* dev_partials, int ld, int rows, int cols, float eps,
                      float* dev_r);
/* Op-level door for the weight preparation of a LayerNorm folded into its consuming linear (the ViT towers' q|k|v and fc1;
 * engine_base.hpp::build_tower): in place, W[n,:] := round(W[n,:] * g - mean_k(W[n,:] * g)) with ZERO-SUM rounding (the rounded
 * row sums to zero within its smallest step, so that the row mean of the activations really drops out of x . W'^T), and
 * bias[n] += W[n,:] . b_ln.  W [n_rows, K] bf16, bias [n_rows] or null, g / b_ln [K]: the LayerNorm's weight / bias.
 * Replaces nn.LayerNorm + nn.Linear of CLIPEncoderLayer / OwlViTEncoderLayer (transformers; reached from
 * VisualSearch/model/llava/model/multimodal_encoder/clip_encoder.py:46-60 and VisualSearch/model/owlvit/owlvit.py:121-126). */
int vstar_op_ln_fold(void* stream, uint16_t* dev_W, uint16_t* dev_bias, const uint16_t* dev_g, const uint16_t* dev_b_ln,
                     int n_rows, int K);
/* Which GEMM kernel the calling thread's last vstar_op_gemm / vstar_op_gemm_fp8 launched: 128, 256, 384 (= 256 + 128: whole
 * rounds of 256x256 tiles over the leading rows and the ragged last round's rows as 128x128 tiles — bit-identical to either
 * kernel alone), or 0 (nothing launched).  Lets a test assert that it exercised the kernel it was written for, whatever the dispatcher's heuristics do. */
int vstar_op_gemm_last_tile(void);
/* Dispatcher dry run (host only, no GPU needed, nothing is launched): which kernel vstar_op_gemm / the engine would pick for
 * C[M,N] = A[M,K] · W^T with this epilogue (+ residual, + the fused-RoPE epilogue of the LLaMA qkv projection) on a device with
 * `cus` compute units.  Returns 10 * tile + variant: tile as vstar_op_gemm_last_tile (128, 256, 384 = split launch, 2560 = the
 * opt-in hand-scheduled kernel), variant of the 128-row family 2 = double buffer, 5 / 6 / 7 = loader-wave ring on the 128 x 128 /
 * 128 x 64 / 128 x 256 tile (0 for the 256^2 kernel; for 384 the variant of the trailing 128-row part); negative = error code. */
int vstar_op_gemm_plan(int M, int N, int K, int epilogue, int has_residual, int fused_rope, int cus);
/* W8A8 GEMM (BASELINE config 5: fp8 weights on the CDNA4 fp8 MFMA), op level, all pointers DEVICE pointers: quantises the
 * rows of A [M,K] (per token) and of W [ceil(N/256)*256, K] (per output channel) to OCP fp8 e4m3 with scale = absmax/448,
 * then C[M,N] = epilogue((A_q . W_q^T) * a_scale[m] * w_scale[n] + bias) (+ residual) with fp32 accumulation on
 * v_mfma_scale_f32_16x16x128_f8f6f4.  Requires M >= 1024, N >= 256, K % 256 == 0; epilogue NONE or SILU_MUL.
 * iters > 0: the GEMM alone is repeated `iters` times between HIP events and *gemm_ms receives the mean (benchmarks). */
int vstar_op_gemm_fp8(void* stream, const uint16_t* dev_A, const uint16_t* dev_W, const uint16_t* dev_bias,
                      const uint16_t* dev_residual, uint16_t* dev_C, int M, int N, int K, int epilogue, int iters, float* gemm_ms);
/* Block-scaled W8A8 (round 6; csrc/mx.hpp): the inputs of o_proj / down_proj as OCP e4m3 bytes with ONE E8M0 byte per row and 32
 * consecutive k (the smallest power of two that brings the block's largest magnitude to <= 448), applied inside
 * v_mfma_scale_f32_16x16x128_f8f6f4 per lane — so the PRODUCERS (attention epilogue, gate|up epilogue) quantise, and the two stand-alone
 * per-token passes of the round-3 scheme disappear.  The engine uses it when llm_w8a8 is set and the step's row count is a multiple of
 * 256 (level 1, the default).  Level 2 (VSTAR_W8A8_MX=2 in the environment): the residual stream ALSO leaves o_proj / down_proj as a
 * block-scaled fp8 copy with sum-of-squares partials, q|k|v and gate|up consume it with their RMSNorm folded (weight into the fp8 W,
 * 1 / rms as the per-row scale) — no activation quantisation pass is left; +1 % speed, but it moved the search on the config-5 leg
 * (another final box), hence opt-in.  VSTAR_W8A8_MX=0: the per-token scheme.  vstar_w8a8_mx_active returns the level the last step
 * ran (0 / 1 / 2).
 * Scale bytes are TILE-MAJOR (vstar_op_mx_scale_offset(row, k / 32, rows)); rows % 128 == 0, cols % 128 == 0.  The reference has no
 * fp8 path: oracle/vsm_oracle.py::mx_fake_quant restates the arithmetic.  Op-level doors (device pointers, synchronous):
 *   vstar_op_quantize_mx     X [rows, cols] bf16 -> q [rows, cols] fp8 + scales (the stand-alone form the fused producers must equal)
 *   vstar_op_gemm_mx         (Aq, a_scales) . quant_per_channel(W)^T, optionally x row_scale[m] (the folded RMSNorm's 1 / rms);
 *                            M % 256 == 0, N % 256 == 0, K % 256 == 0.  epilogue NONE: C [M, N] bf16 (+ residual); with dev_C8 also the
 *                            block-scaled fp8 copy of the stored rows (dev_c_scales) and, if dev_sumsq, their sums of squares per 64
 *                            columns [M, N / 64] — same bytes as vstar_op_quantize_mx over C.  epilogue SILU_MUL (W rows interleaved
 *                            gate|up 16 | 16): [M, N / 2] as bf16 (dev_C8 null) or as fp8 + scales (dev_C8: dev_C unused)
 *   vstar_op_gemm_fp8_mxout  per-token-quantised A . W^T with W rows interleaved gate|up (16 | 16), SiLU(gate) * up written as fp8 + scales
 *                            [M, N / 2] — same bytes as vstar_op_gemm_fp8(..., VSTAR_EPI_SILU_MUL) followed by vstar_op_quantize_mx
 *                            (iters / gemm_ms as in vstar_op_gemm_fp8)
 *   vstar_op_attention_mx    causal D = 128 attention on a fused qkv buffer, output [B*S, H*128] as fp8 + scales — same bytes as
 *                            vstar_op_attention(rope_theta = 0) followed by vstar_op_quantize_mx */
size_t vstar_op_mx_scale_bytes(int rows, int cols);
int64_t vstar_op_mx_scale_offset(int row, int k_block, int rows);
int vstar_op_quantize_mx(void* stream, const uint16_t* dev_X, uint8_t* dev_q, uint8_t* dev_scales, int rows, int cols);
int vstar_op_gemm_mx(void* stream, const uint8_t* dev_Aq, const uint8_t* dev_a_scales, const float* dev_row_scale, const uint16_t* dev_W,
                     const uint16_t* dev_residual, uint16_t* dev_C, uint8_t* dev_C8, uint8_t* dev_c_scales, float* dev_sumsq, int M, int N, int K,
                     int epilogue, int iters, float* gemm_ms);
int vstar_op_gemm_fp8_mxout(void* stream, const uint16_t* dev_A, const uint16_t* dev_W, uint8_t* dev_C8, uint8_t* dev_c_scales, int M, int N,
                            int K, int iters, float* gemm_ms);
int vstar_op_attention_mx(void* stream, const uint16_t* dev_qkv, uint8_t* dev_out8, uint8_t* dev_scales, int B, int S, int H);
int vstar_w8a8_mx_active(vstar_handle* h);
/* LayerNorm over the last dim (eps, affine) / LLaMA RMSNorm.  bf16 in/out. */
int vstar_op_layernorm(void* stream, const uint16_t* dev_x, const uint16_t* dev_gamma, const uint16_t* dev_beta,
                       uint16_t* dev_y, int rows, int cols, float eps);
int vstar_op_rmsnorm(void* stream, const uint16_t* dev_x, const uint16_t* dev_gamma, uint16_t* dev_y,
                     int rows, int cols, float eps);
/* Multi-head attention softmax(Q K^T * scale [+causal]) V on a fused [B*S, 3*H*D] qkv buffer; D in {64,128}.
 * If rope_theta > 0, rotate-half RoPE (positions 0..S-1) is applied to q and k first (LLaMA). Output [B*S, H*D]. */
int vstar_op_attention(void* stream, uint16_t* dev_qkv, uint16_t* dev_out, void* dev_workspace, size_t workspace_bytes,
                       int B, int S, int H, int D, int causal, float rope_theta);
size_t vstar_op_attention_workspace(int B, int S, int H, int D);

#ifdef __cplusplus
}
#endif
#endif /* VSTAR_HIP_H */
