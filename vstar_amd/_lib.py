"""ctypes binding of libvstar_hip.so (the C-ABI declared in include/vstar_hip.h).

There is deliberately no CPU / PyTorch fallback: if the HIP library is missing or no GPU is visible, every compute
entry point raises.  Build with `python -c "import __graft_entry__ as g; g.build()"` or `vstar_amd/csrc/build.sh`.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_uint, c_uint16, c_void_p

from .config import CVqaConfig, CVstarConfig, MASK_RES, MAX_VERIFY, N_BOXES

LIB_PATH = os.environ.get("VSTAR_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libvstar_hip.so")   # VSTAR_LIB: A/B builds

# every symbol include/vstar_hip.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "vstar_create", "vstar_destroy", "vstar_last_error", "vstar_load_tensor", "vstar_finalize_weights",
    "vstar_vsm_score_batch", "vstar_upsample_mask", "vstar_debug_read", "vstar_stream", "vstar_profile_enable",
    "vstar_profile_read", "vstar_profile_read_fp8", "vstar_op_gemm", "vstar_op_layernorm", "vstar_op_rmsnorm", "vstar_op_attention",
    "vstar_op_attention_workspace", "vstar_image_set", "vstar_preprocess_crops", "vstar_heatmap_stats", "vstar_heatmap_stats_batch", "vstar_vsm_generate", "vstar_op_gemm_fp8",
    "vstar_op_gemm_last_tile", "vstar_op_gemm_plan", "vstar_op_gemm_norm", "vstar_op_rms_rstd", "vstar_op_ln_fold", "vstar_upsample_mask_ex", "vstar_vsm_score_grouped",
    "vstar_image_set_slot", "vstar_image_set_slot_async", "vstar_preprocess_crops_slots", "vstar_comm_unique_id", "vstar_comm_init", "vstar_allgather_results",
    "vstar_comm_destroy", "vstar_build_source_hash", "vstar_op_mx_scale_bytes", "vstar_op_mx_scale_offset", "vstar_op_quantize_mx",
    "vstar_op_gemm_mx", "vstar_op_gemm_fp8_mxout", "vstar_op_attention_mx", "vstar_w8a8_mx_active",
]

# every symbol include/vstar_vqa.h declares
EXPORTS_VQA = [
    "vstar_vqa_create", "vstar_vqa_destroy", "vstar_vqa_last_error", "vstar_vqa_load_tensor", "vstar_vqa_finalize_weights",
    "vstar_vqa_encode_images", "vstar_vqa_forward", "vstar_vqa_debug_read", "vstar_vqa_last_forward_ms", "vstar_vqa_op_gemm",
]

F32, F16, BF16 = 0, 1, 2
MAX_IMAGE_SLOTS = 64
EPI_NONE, EPI_QUICK_GELU, EPI_GELU, EPI_RELU, EPI_SILU_MUL = range(5)
EPI_NOSYNC, EPI_TILE128, EPI_TILE256, EPI_TILE4W = 0x100, 0x200, 0x400, 0x800
TILE_4W = 2564          # vstar_op_gemm_last_tile() value of the 4-wave / AGPR 256 x 256 kernel
F_SKIP_OWL, F_DEVICE_INPUTS, F_DEVICE_OUTPUT, F_NO_SYNC, F_INTERNAL_PIXELS, F_SHARE_PREFIX = 1, 2, 4, 8, 16, 32


class VstarResult(ctypes.Structure):
    _fields_ = [
        ("pred_logits", c_float * N_BOXES),
        ("pred_boxes", c_float * (N_BOXES * 4)),
        ("lowres_mask", c_float * (MASK_RES * MASK_RES)),
        ("tf_argmax", c_int32 * MAX_VERIFY),
    ]


RESULT_FLOATS = ctypes.sizeof(VstarResult) // 4


class VstarError(RuntimeError):
    pass


_lib = None


def load() -> ctypes.CDLL:
    """Loads the shared library and declares the prototypes.  Raises (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VstarError(f"{LIB_PATH} not built — run __graft_entry__.build() (hipcc --offload-arch=gfx950); "
                         "there is no CPU fallback for the HIP engine")
    # torch first: the process must end up with ONE HIP runtime (torch bundles its own libamdhip64; loading ours against
    # /opt/rocm before torch leaves the engine on a runtime that sees no device once torch is imported)
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    H = c_void_p
    lib.vstar_create.argtypes = [POINTER(CVstarConfig), c_int, POINTER(H)]
    lib.vstar_create.restype = c_int
    lib.vstar_destroy.argtypes = [H]
    lib.vstar_destroy.restype = None
    lib.vstar_last_error.argtypes = [H]
    lib.vstar_last_error.restype = c_char_p
    lib.vstar_build_source_hash.argtypes = []
    lib.vstar_build_source_hash.restype = c_char_p
    lib.vstar_load_tensor.argtypes = [H, c_char_p, c_void_p, c_int, c_int, POINTER(c_int64)]
    lib.vstar_load_tensor.restype = c_int
    lib.vstar_finalize_weights.argtypes = [H]
    lib.vstar_finalize_weights.restype = c_int
    lib.vstar_vsm_score_batch.argtypes = [H, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_uint,
                                          c_void_p]
    lib.vstar_vsm_score_batch.restype = c_int
    lib.vstar_vsm_score_grouped.argtypes = [H, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int,
                                            c_uint, c_void_p]
    lib.vstar_vsm_score_grouped.restype = c_int
    lib.vstar_vsm_generate.argtypes = [H, c_void_p, c_void_p, c_int, c_int, c_int, c_uint, c_void_p, c_void_p]
    lib.vstar_vsm_generate.restype = c_int
    lib.vstar_image_set.argtypes = [H, c_void_p, c_int, c_int]
    lib.vstar_image_set.restype = c_int
    lib.vstar_preprocess_crops.argtypes = [H, c_int, c_void_p]
    lib.vstar_preprocess_crops.restype = c_int
    lib.vstar_image_set_slot.argtypes = [H, c_int, c_void_p, c_int, c_int]
    lib.vstar_image_set_slot.restype = c_int
    lib.vstar_image_set_slot_async.argtypes = [H, c_int, c_void_p, c_int, c_int]
    lib.vstar_image_set_slot_async.restype = c_int
    lib.vstar_preprocess_crops_slots.argtypes = [H, c_int, c_void_p, c_void_p]
    lib.vstar_preprocess_crops_slots.restype = c_int
    lib.vstar_comm_unique_id.argtypes = [c_void_p]
    lib.vstar_comm_unique_id.restype = c_int
    lib.vstar_comm_init.argtypes = [H, c_void_p, c_int, c_int]
    lib.vstar_comm_init.restype = c_int
    lib.vstar_allgather_results.argtypes = [H, c_void_p, c_int, c_void_p, ctypes.c_uint]
    lib.vstar_allgather_results.restype = c_int
    lib.vstar_comm_destroy.argtypes = [H]
    lib.vstar_comm_destroy.restype = c_int
    lib.vstar_heatmap_stats.argtypes = [H, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]
    lib.vstar_heatmap_stats.restype = c_int
    lib.vstar_heatmap_stats_batch.argtypes = [H, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.vstar_heatmap_stats_batch.restype = c_int
    lib.vstar_upsample_mask.argtypes = [H, c_void_p, c_int, c_int, c_void_p]
    lib.vstar_upsample_mask.restype = c_int
    lib.vstar_upsample_mask_ex.argtypes = [H, c_void_p, c_int, c_int, c_int, c_void_p]
    lib.vstar_upsample_mask_ex.restype = c_int
    lib.vstar_debug_read.argtypes = [H, c_char_p, c_void_p, c_int64]
    lib.vstar_debug_read.restype = c_int64
    lib.vstar_stream.argtypes = [H]
    lib.vstar_stream.restype = c_void_p
    lib.vstar_profile_enable.argtypes = [H, c_int]
    lib.vstar_profile_enable.restype = c_int
    lib.vstar_profile_read.argtypes = [H, POINTER(c_double), POINTER(c_int64), POINTER(c_double)]
    lib.vstar_profile_read.restype = c_int
    lib.vstar_profile_read_fp8.argtypes = [H, POINTER(c_double), POINTER(c_int64), POINTER(c_double)]
    lib.vstar_profile_read_fp8.restype = c_int
    lib.vstar_op_gemm.argtypes = [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                                  c_int, c_int, c_int, c_int, c_int]
    lib.vstar_op_gemm.restype = c_int
    lib.vstar_op_gemm_norm.argtypes = [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int,
                                       c_int, c_void_p, c_void_p, c_int]
    lib.vstar_op_gemm_norm.restype = c_int
    lib.vstar_op_rms_rstd.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p]
    lib.vstar_op_rms_rstd.restype = c_int
    lib.vstar_op_ln_fold.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int]
    lib.vstar_op_ln_fold.restype = c_int
    lib.vstar_op_gemm_last_tile.argtypes = []
    lib.vstar_op_gemm_last_tile.restype = c_int
    lib.vstar_op_gemm_plan.argtypes = [c_int] * 7
    lib.vstar_op_gemm_plan.restype = c_int
    lib.vstar_op_gemm_fp8.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                      POINTER(c_float)]
    lib.vstar_op_gemm_fp8.restype = c_int
    lib.vstar_op_mx_scale_bytes.argtypes = [c_int, c_int]
    lib.vstar_op_mx_scale_bytes.restype = c_size_t
    lib.vstar_op_mx_scale_offset.argtypes = [c_int, c_int, c_int]
    lib.vstar_op_mx_scale_offset.restype = ctypes.c_int64
    lib.vstar_op_quantize_mx.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int]
    lib.vstar_op_quantize_mx.restype = c_int
    lib.vstar_op_gemm_mx.argtypes = [c_void_p] * 10 + [c_int, c_int, c_int, c_int, c_int, POINTER(c_float)]
    lib.vstar_op_gemm_mx.restype = c_int
    lib.vstar_op_gemm_fp8_mxout.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, POINTER(c_float)]
    lib.vstar_op_gemm_fp8_mxout.restype = c_int
    lib.vstar_op_attention_mx.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int]
    lib.vstar_op_attention_mx.restype = c_int
    lib.vstar_w8a8_mx_active.argtypes = [c_void_p]
    lib.vstar_w8a8_mx_active.restype = c_int
    lib.vstar_op_layernorm.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float]
    lib.vstar_op_layernorm.restype = c_int
    lib.vstar_op_rmsnorm.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float]
    lib.vstar_op_rmsnorm.restype = c_int
    lib.vstar_op_attention.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_int, c_int, c_int,
                                       c_float]
    lib.vstar_op_attention.restype = c_int
    lib.vstar_op_attention_workspace.argtypes = [c_int, c_int, c_int, c_int]
    lib.vstar_op_attention_workspace.restype = c_size_t
    # ---- VQA-LLM engine (include/vstar_vqa.h) ----
    lib.vstar_vqa_create.argtypes = [POINTER(CVqaConfig), c_int, POINTER(H)]
    lib.vstar_vqa_create.restype = c_int
    lib.vstar_vqa_destroy.argtypes = [H]
    lib.vstar_vqa_destroy.restype = None
    lib.vstar_vqa_last_error.argtypes = [H]
    lib.vstar_vqa_last_error.restype = c_char_p
    lib.vstar_vqa_load_tensor.argtypes = [H, c_char_p, c_void_p, c_int, c_int, POINTER(c_int64)]
    lib.vstar_vqa_load_tensor.restype = c_int
    lib.vstar_vqa_finalize_weights.argtypes = [H]
    lib.vstar_vqa_finalize_weights.restype = c_int
    lib.vstar_vqa_encode_images.argtypes = [H, c_int, c_void_p, c_int]
    lib.vstar_vqa_encode_images.restype = c_int
    lib.vstar_vqa_forward.argtypes = [H, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                      c_void_p]
    lib.vstar_vqa_forward.restype = c_int
    lib.vstar_vqa_op_gemm.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                      c_void_p, c_float]
    lib.vstar_vqa_op_gemm.restype = c_int
    lib.vstar_vqa_debug_read.argtypes = [H, c_char_p, c_void_p, c_int64]
    lib.vstar_vqa_debug_read.restype = c_int64
    lib.vstar_vqa_last_forward_ms.argtypes = [H]
    lib.vstar_vqa_last_forward_ms.restype = c_double
    _lib = lib
    return lib


def check_vqa(rc: int, handle=None) -> None:
    if rc != 0:
        msg = load().vstar_vqa_last_error(handle)
        raise VstarError(f"libvstar_hip (vqa) error {rc}: {msg.decode() if msg else '?'}")


def check(rc: int, handle=None) -> None:
    if rc != 0:
        msg = load().vstar_last_error(handle)
        raise VstarError(f"libvstar_hip error {rc}: {msg.decode() if msg else '?'}")
