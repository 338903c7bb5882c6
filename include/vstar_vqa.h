/*
 * vstar_vqa.h — C-ABI of the VQA-LLM engine inside libvstar_hip.so (MI355X / gfx950, IEEE fp16 storage, fp32 accumulate):
 * the second model of the V* pipeline, i.e. `LlavaSearchLlamaForCausalLM`
 *   CLIP-ViT-L/14@224 -> { mm_projector ("long", 256 tokens/image) | mm_projector_object = LayerNorm + PerceiverResampler
 *   + Linear ("short", 32 tokens/image) } -> <image>/<object> splice -> LLaMA-7B with a KV cache -> lm_head,
 * as driven by VQA_LLM.free_form_inference / multiple_choices_inference (vstar_bench_eval.py:78-165).
 *
 * The reference has no FFI; the Python call sites each entry point replaces are cited below (paths relative to the
 * reference repo root).  Conventions are those of vstar_hip.h (0 = OK, negative = error, caller owns host buffers, one
 * handle per device, stream-ordered, synchronised before return).  fp16 tensors cross the ABI as uint16_t bits.
 *
 * Design (MI355X-first, not a port of HF generate): the KV cache is a set of `max_slots` fixed slots of `max_ctx`
 * positions ([layer][slot][head][ctx][128], K and V) resident in HBM; a forward call advances any number of sequences by
 * any number of new rows each (prefill, one-token decode steps of many sequences at once, or teacher-forced option
 * continuations), and a sequence may read its first `past_len` positions from ANOTHER slot — the multiple-choice options
 * fork the question's cache without copying it.  Rows are vocabulary ids or rows of the device-resident feature table
 * written by vstar_vqa_encode_images, so image/object features never leave the GPU.
 */
#ifndef VSTAR_VQA_H
#define VSTAR_VQA_H

#include "vstar_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define VSTAR_VQA_ABI_VERSION 1
#define VSTAR_VQA_PAD_ROW INT32_MIN      /* a row source meaning "zero embedding" */

/* Geometry: LlavaSearchConfig(LlamaConfig) + CLIP tower + projector builder
 * (LLaVA/llava/model/language_model/llava_search_llama.py:30-50, multimodal_projector/builder.py:33-68). */
typedef struct vstar_vqa_config {
  int32_t abi_version;        /* VSTAR_VQA_ABI_VERSION */
  int32_t clip_image_size;    /* 224 */
  int32_t clip_patch;         /* 14 */
  int32_t clip_hidden;        /* 1024 */
  int32_t clip_heads;         /* 16 */
  int32_t clip_mlp;           /* 4096 */
  int32_t clip_layers;        /* 24 */
  int32_t clip_select_layer;  /* -2, patch tokens only (mm_vision_select_feature = 'patch') */
  int32_t llm_hidden;         /* 4096 */
  int32_t llm_heads;          /* 32 (head dim 128) */
  int32_t llm_mlp;            /* 11008 */
  int32_t llm_layers;         /* 32 */
  int32_t llm_vocab;          /* len(tokenizer) after builder.py:131-135 */
  float   llm_rms_eps;
  float   llm_rope_theta;
  int32_t projector_type;     /* mm_projector: 0 = linear, 1 = mlp2x_gelu (builder.py:39-49) */
  int32_t pcv_depth;          /* 6   PerceiverResampler(depth, heads, dim_head, num_latents), builder.py:54-66 */
  int32_t pcv_heads;          /* 16 */
  int32_t pcv_dim_head;       /* 96 */
  int32_t pcv_latents;        /* 32 = short tokens per image */
  int32_t pcv_ff_mult;        /* 4 */
  int32_t max_slots;          /* KV-cache slots */
  int32_t max_ctx;            /* positions per slot */
  int32_t max_rows;           /* new rows per forward call (after padding a ragged prefill batch) */
  int32_t max_images;         /* feature-table slots; each holds P long rows then pcv_latents short rows */
  int32_t reserved[8];
} vstar_vqa_config;

typedef struct vstar_vqa_engine vstar_vqa_handle;

/* Replaces load_pretrained_model(...) (LLaVA/llava/model/builder.py:26-151, called at vstar_bench_eval.py:46). */
int vstar_vqa_create(const vstar_vqa_config* cfg, int device, vstar_vqa_handle** out);
void vstar_vqa_destroy(vstar_vqa_handle* h);
const char* vstar_vqa_last_error(const vstar_vqa_handle* h);
/* HF state-dict keys of LlavaSearchLlamaForCausalLM ("model.layers.N...", "model.mm_projector...",
 * "model.mm_projector_object.{0,1,2}...", "lm_head.weight") plus the separately loaded CLIP tower under "clip.". */
int vstar_vqa_load_tensor(vstar_vqa_handle* h, const char* key, const void* host_ptr, int dtype, int ndim,
                          const int64_t* shape);
int vstar_vqa_finalize_weights(vstar_vqa_handle* h);

/* encode_images / project_features (LLaVA/llava/model/llava_search_arch.py:84-94): CLIP tower, then BOTH projectors, for
 * n images given as fp16 pixels [n,3,I,I] (CLIPImageProcessor output, .half()).  Image i fills feature slot
 * first_slot + i: rows [0,P) = long features, rows [P, P+pcv_latents) = short features; a feature row's global index is
 * slot * (P + pcv_latents) + row, and a forward row source of -(1 + index) splices it
 * (prepare_inputs_labels_for_multimodal, llava_search_arch.py:96-266). */
int vstar_vqa_encode_images(vstar_vqa_handle* h, int n, const uint16_t* pixels_f16, int first_slot);

/* One model forward over new rows of nseq sequences — LlavaSearchLlamaForCausalLM.forward
 * (llava_search_llama.py:56-113) for the prefill (vstar_bench_eval.py:127-133), each generate() step (:90-103) and each
 * option continuation with past_key_values (:148-151).
 *   row_off[nseq+1]  sequence i contributes rows row_off[i] .. row_off[i+1]-1 of src
 *   src[rows]        >= 0: vocabulary id; < 0: feature row -(1+index); VSTAR_VQA_PAD_ROW: zero row
 *   kv_slot[nseq]    slot that receives the new rows' K/V at positions past_len[i], past_len[i]+1, ...
 *   prefix_slot[nseq] slot holding positions [0, past_len[i]) (== kv_slot[i] unless the sequence forks a shared prefix)
 *   past_len[nseq]   number of cached positions in front of the new rows
 *   want[n_want]     indices into src of the rows whose logits are needed (lm_head runs on these rows only)
 *   logits_f16       [n_want, llm_vocab] fp16 logits (may be NULL);  argmax [n_want] (may be NULL) */
int vstar_vqa_forward(vstar_vqa_handle* h, int nseq, const int32_t* row_off, const int32_t* src, const int32_t* kv_slot,
                      const int32_t* prefix_slot, const int32_t* past_len, int n_want, const int32_t* want,
                      uint16_t* logits_f16, int32_t* argmax);

/* Op-level entry for tests and micro-benchmarks, fp16, all pointers DEVICE pointers: C[M,N] = epilogue(A[M,K] · W[N,K]^T
 * + bias) (+ residual), epilogue codes and operand rules as vstar_op_gemm (W rows padded to a multiple of 256, K % 64 == 0).
 * kernel: 0 = the engine's dispatch (weight-streaming kernel for M <= 64, MFMA tile kernels otherwise), 1 = force the
 * weight-streaming kernel (M <= 64; for M <= 8 that is its LDS-ring variant), 2 = force the tile kernels, 3 = the weight-streaming
 * kernel with the weights through registers even where the LDS-ring variant would run (bit-identity tests), 4 = the 4-wave / AGPR
 * 256x256 tile kernel (gemm4w, round 6; error outside its domain), 5 = the 8-wave 256x256 kernel.  dev_norm_w (nullable, weight-streaming kernel only):
 * LlamaRMSNorm gains [K]; the rows of A are RMS-normalised (eps = norm_eps) while they are loaded, as the decode path
 * does for input_layernorm -> q/k/v and post_attention_layernorm -> gate/up.  Runs on the null stream and synchronises. */
int vstar_vqa_op_gemm(const void* dev_A, const void* dev_W, const void* dev_bias, const void* dev_residual, void* dev_C,
                      int M, int N, int K, int epilogue, int kernel, const void* dev_norm_w, float norm_eps);

/* Diagnostics for the parity tests: "features" = the whole feature table, fp16 -> float. Returns elements written. */
int64_t vstar_vqa_debug_read(vstar_vqa_handle* h, const char* name, float* out, int64_t capacity);
/* Timing of the last forward call in milliseconds (HIP events on the engine stream). */
double vstar_vqa_last_forward_ms(const vstar_vqa_handle* h);

#ifdef __cplusplus
}
#endif
#endif /* VSTAR_VQA_H */
