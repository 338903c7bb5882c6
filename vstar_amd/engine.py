"""Python face of the HIP engine: weight hand-over and the batched per-crop scoring call.

`VstarEngine.score_batch` is the batched, single-prefill equivalent of `VSMForCausalLM.inference(...)` /
`model_forward(inference=True)` (VisualSearch/model/VSM.py:438-553, 201-364).  All arithmetic happens in
libvstar_hip.so; torch is used here only to hold host/device buffers.
"""
from __future__ import annotations

import ctypes
from typing import Dict, List, Optional

import numpy as np
import torch

from . import _lib
from .config import IMAGE_TOKEN_INDEX, MASK_RES, MAX_VERIFY, N_BOXES, VSMConfig
from .weights import dense_pe, state_dict_spec

_DT = {torch.float32: _lib.F32, torch.float16: _lib.F16, torch.bfloat16: _lib.BF16}


def _as_bf16(t) -> torch.Tensor:
    if isinstance(t, np.ndarray):
        t = torch.from_numpy(t)
    if t.dtype != torch.bfloat16:
        t = t.to(torch.bfloat16)
    return t.contiguous()


class VstarEngine:
    def __init__(self, cfg: VSMConfig, device: int = 0):
        self.cfg = cfg
        self.lib = _lib.load()
        self.handle = ctypes.c_void_p()
        c = cfg.to_c()
        _lib.check(self.lib.vstar_create(ctypes.byref(c), device, ctypes.byref(self.handle)))
        self.device = device
        self.finalized = False
        # VSTAR_F_SHARE_PREFIX on score_batch / score_boxes calls (the system prompt shared by the crops of a call is computed
        # once).  Off by default: at 32 crops on 256 CUs the 5 % fewer GEMM rows (76 instead of 80 row tiles) fill the same number of
        # whole rounds of 256^2 tiles, so the call is no faster (tools/prefix_bench.py: 233.6 vs 233.5 ms); it pays when the saved
        # rows cross a round boundary (e.g. 33 crops then cost what 32 do)
        self.share_prefix = False

    # ---- weights (replaces VSMForCausalLM.from_pretrained + get_vision_tower(), visual_search.py:157-161) ----
    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True) -> None:
        spec = state_dict_spec(self.cfg)
        needed = [k for k in spec if not k.endswith("gaussian_matrix")]
        for k in needed:
            if k not in sd:
                if strict:
                    raise KeyError(f"checkpoint tensor missing: {k}")
                continue
            t = sd[k]
            # layers of the CLIP tower beyond hidden_states[select_layer] are never executed
            if k.startswith("clip.vision_model.encoder.layers."):
                if int(k.split(".")[4]) >= self.cfg.clip_blocks:
                    continue
            if k.startswith("clip.vision_model.post_layernorm"):
                continue
            self._load(k, t)
        gm = sd["model.prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"]
        self._load("sam.dense_pe", dense_pe(gm.cpu()))
        _lib.check(self.lib.vstar_finalize_weights(self.handle), self.handle)
        self.finalized = True

    def _load(self, key: str, t: torch.Tensor) -> None:
        t = t.detach().cpu().contiguous()
        if t.dtype not in _DT:
            t = t.float()
        shape = (ctypes.c_int64 * max(t.dim(), 1))(*t.shape)
        _lib.check(self.lib.vstar_load_tensor(self.handle, key.encode(), ctypes.c_void_p(t.data_ptr()), _DT[t.dtype], t.dim(),
                                              shape), self.handle)

    # ---- free-text greedy decode with a KV cache (VSM.inference mode='vqa', VSM.py:438-462) ----
    def generate(self, clip_pix, input_ids, max_new_tokens: int = 100, eos_id: int = 2):
        """clip_pix [1,3,I,I] (bf16-castable, CPU), input_ids [L] with one -200 -> list of generated ids (EOS included)."""
        clip_pix = _as_bf16(clip_pix).cpu()
        ids = np.ascontiguousarray(np.asarray(input_ids, np.int32).reshape(-1))
        out = np.zeros((max_new_tokens,), np.int32)
        n = ctypes.c_int32(0)
        _lib.check(self.lib.vstar_vsm_generate(self.handle, ctypes.c_void_p(clip_pix.data_ptr()), ctypes.c_void_p(ids.ctypes.data),
                                               ids.shape[0], max_new_tokens, eos_id, 0, ctypes.c_void_p(out.ctypes.data),
                                               ctypes.byref(n)), self.handle)
        return out[:n.value].tolist()

    # ---- the hot path ----
    def score_batch(self, clip_pix, owl_pix, input_ids, loc_pos, verify_pos=None, skip_owl: bool = False,
                    sync: bool = True, raw: bool = False, out_dev: Optional[torch.Tensor] = None, share_prefix: Optional[bool] = None):
        """clip_pix [B,3,I,I], owl_pix [B,3,768,768] (bf16-castable; CPU or cuda tensors), input_ids [B,L] with one -200,
        loc_pos [B] spliced-sequence index of the hidden state that predicts [LOC]; verify_pos [B,V] optional.
        out_dev: a contiguous float32 cuda tensor [B, RESULT_FLOATS] — the records stay in HBM (VSTAR_F_DEVICE_OUTPUT: what the
        multi-GPU path all-gathers over RCCL without a host bounce); the call then returns None.
        share_prefix (default: the engine's `share_prefix` attribute, False): VSTAR_F_SHARE_PREFIX — the text before <image>, when it
        is the same >= 16 tokens in every row (the system prompt), goes through LLaMA once instead of B times."""
        cfg = self.cfg
        clip_pix = _as_bf16(clip_pix)
        B = clip_pix.shape[0]
        I = cfg.clip_image_size
        assert tuple(clip_pix.shape) == (B, 3, I, I), clip_pix.shape
        flags = _lib.F_SHARE_PREFIX if (self.share_prefix if share_prefix is None else share_prefix) else 0
        if skip_owl:
            flags |= _lib.F_SKIP_OWL
            owl_ptr = None
        else:
            owl_pix = _as_bf16(owl_pix)
            assert tuple(owl_pix.shape) == (B, 3, cfg.owl_image_size, cfg.owl_image_size), owl_pix.shape
            assert owl_pix.device == clip_pix.device
            owl_ptr = ctypes.c_void_p(owl_pix.data_ptr())
        if clip_pix.is_cuda:
            flags |= _lib.F_DEVICE_INPUTS
        if not sync:
            flags |= _lib.F_NO_SYNC
        ids = np.ascontiguousarray(np.asarray(input_ids, dtype=np.int32))
        assert ids.ndim == 2 and ids.shape[0] == B
        loc = np.ascontiguousarray(np.asarray(loc_pos, dtype=np.int32))
        nv = 0
        vptr = None
        if verify_pos is not None:
            vp = np.ascontiguousarray(np.asarray(verify_pos, dtype=np.int32)).reshape(B, -1)
            nv = vp.shape[1]
            assert nv <= MAX_VERIFY
            vptr = vp.ctypes.data_as(ctypes.c_void_p)
        out, out_ptr = self._out_buffer(B, out_dev)
        if out_dev is not None:
            flags |= _lib.F_DEVICE_OUTPUT
        self._keep = (clip_pix, owl_pix, ids, loc, out)  # keep alive for no-sync calls
        _lib.check(self.lib.vstar_vsm_score_batch(
            self.handle, B, ctypes.c_void_p(clip_pix.data_ptr()), owl_ptr, ids.ctypes.data_as(ctypes.c_void_p),
            ids.shape[1], loc.ctypes.data_as(ctypes.c_void_p), vptr, nv, flags, out_ptr), self.handle)
        if not sync or out_dev is not None:
            return None
        return out if raw else self.unpack(out, nv)

    @staticmethod
    def _out_buffer(B: int, out_dev: Optional[torch.Tensor]):
        if out_dev is None:
            out = np.empty((B, _lib.RESULT_FLOATS), dtype=np.float32)
            return out, out.ctypes.data_as(ctypes.c_void_p)
        assert out_dev.is_cuda and out_dev.dtype == torch.float32 and out_dev.is_contiguous(), "out_dev: contiguous fp32 cuda tensor"
        assert tuple(out_dev.shape) == (B, _lib.RESULT_FLOATS), out_dev.shape
        return out_dev, ctypes.c_void_p(out_dev.data_ptr())

    # ---- grouped scoring: G crops x T prompts sharing their first Lp ids (include/vstar_hip.h: vstar_vsm_score_grouped) ----
    def score_grouped(self, clip_pix, owl_pix, prefix_ids, suffix_ids, loc_in_suffix, verify_in_suffix=None, raw: bool = False,
                      internal_pixels: bool = False, n_crops: Optional[int] = None):
        """prefix_ids [Lp] (one -200); suffix_ids [G, T, Ls] (Ls <= 32, right-padded); loc_in_suffix [G, T]; verify_in_suffix
        [G, T, V] or None.  Pixels of the G crops as for score_batch, or internal_pixels=True after preprocess (score_boxes path).
        Returns G*T records in (crop-major, prompt-minor) order."""
        suf = np.ascontiguousarray(np.asarray(suffix_ids, dtype=np.int32))
        assert suf.ndim == 3
        G, T, Ls = suf.shape
        pre = np.ascontiguousarray(np.asarray(prefix_ids, dtype=np.int32).reshape(-1))
        loc = np.ascontiguousarray(np.asarray(loc_in_suffix, dtype=np.int32).reshape(G * T))
        nv, vptr = 0, None
        if verify_in_suffix is not None:
            vp = np.ascontiguousarray(np.asarray(verify_in_suffix, dtype=np.int32)).reshape(G * T, -1)
            nv = vp.shape[1]
            vptr = vp.ctypes.data_as(ctypes.c_void_p)
        flags = 0
        cptr = optr = None
        if internal_pixels:
            flags |= _lib.F_INTERNAL_PIXELS
        else:
            clip_pix, owl_pix = _as_bf16(clip_pix), _as_bf16(owl_pix)
            assert clip_pix.shape[0] == G and owl_pix.shape[0] == G and clip_pix.device == owl_pix.device
            if clip_pix.is_cuda:
                flags |= _lib.F_DEVICE_INPUTS
            cptr, optr = ctypes.c_void_p(clip_pix.data_ptr()), ctypes.c_void_p(owl_pix.data_ptr())
        out = np.empty((G * T, _lib.RESULT_FLOATS), dtype=np.float32)
        _lib.check(self.lib.vstar_vsm_score_grouped(
            self.handle, G, T, cptr, optr, pre.ctypes.data_as(ctypes.c_void_p), pre.shape[0], suf.ctypes.data_as(ctypes.c_void_p), Ls,
            loc.ctypes.data_as(ctypes.c_void_p), vptr, nv, flags, out.ctypes.data_as(ctypes.c_void_p)), self.handle)
        return out if raw else self.unpack(out, nv)

    # ---- GPU-side preprocessing ----
    def set_image(self, image, slot: int = 0) -> None:
        """Uploads the full RGB image (PIL.Image or uint8 [H,W,3]) once into image slot `slot` (0 .. _lib.MAX_IMAGE_SLOTS-1);
        crops are then just (slot, box) pairs, and one batch may mix crops of different resident images."""
        # (an RGB PIL image needs no convert(): that alone is a 25 MB copy + 10-30 ms per 4K image in front of every upload)
        if hasattr(image, "convert") and getattr(image, "mode", None) != "RGB":
            image = image.convert("RGB")
        arr = np.ascontiguousarray(np.asarray(image, dtype=np.uint8))
        assert arr.ndim == 3 and arr.shape[2] == 3
        self._image_hw = arr.shape[:2]
        _lib.check(self.lib.vstar_image_set_slot(self.handle, int(slot), arr.ctypes.data_as(ctypes.c_void_p), arr.shape[0], arr.shape[1]),
                   self.handle)

    def set_image_async(self, image, slot: int = 0) -> None:
        """set_image without stalling the scoring stream (vstar_image_set_slot_async): the pixels are staged in pinned memory before
        this returns and travel on a copy stream; the next preprocessing of the slot waits for them on the device.  Thread-safe
        against a scoring call in progress on another thread — the stream search calls it from its prefetch thread."""
        if hasattr(image, "convert") and getattr(image, "mode", None) != "RGB":
            image = image.convert("RGB")
        arr = np.ascontiguousarray(np.asarray(image, dtype=np.uint8))
        assert arr.ndim == 3 and arr.shape[2] == 3
        _lib.check(self.lib.vstar_image_set_slot_async(self.handle, int(slot), arr.ctypes.data_as(ctypes.c_void_p), arr.shape[0], arr.shape[1]),
                   self.handle)

    def _preprocess(self, boxes: np.ndarray, slots) -> None:
        sp = None
        if slots is not None:
            sl = np.ascontiguousarray(np.asarray(slots, dtype=np.int32)).reshape(-1)
            assert sl.shape[0] == boxes.shape[0]
            sp = sl.ctypes.data_as(ctypes.c_void_p)
        _lib.check(self.lib.vstar_preprocess_crops_slots(self.handle, boxes.shape[0], boxes.ctypes.data_as(ctypes.c_void_p), sp), self.handle)

    def score_boxes(self, boxes_xyxy, input_ids, loc_pos, verify_pos=None, raw: bool = False,
                    out_dev: Optional[torch.Tensor] = None, share_prefix: Optional[bool] = None, slots=None):
        """Crop + pad + PIL-exact resize + normalise on the GPU for `boxes_xyxy` [B,4] (ints, as passed to image.crop) of the
        resident image(s) (`slots` [B]: the image slot of each box; None = slot 0), then the same scoring pass as `score_batch`
        (out_dev: see there)."""
        boxes = np.ascontiguousarray(np.asarray(boxes_xyxy, dtype=np.int32)).reshape(-1, 4)
        B = boxes.shape[0]
        self._preprocess(boxes, slots)
        ids = np.ascontiguousarray(np.asarray(input_ids, dtype=np.int32))
        loc = np.ascontiguousarray(np.asarray(loc_pos, dtype=np.int32))
        nv, vptr = 0, None
        if verify_pos is not None:
            vp = np.ascontiguousarray(np.asarray(verify_pos, dtype=np.int32)).reshape(B, -1)
            nv = vp.shape[1]
            vptr = vp.ctypes.data_as(ctypes.c_void_p)
        out, out_ptr = self._out_buffer(B, out_dev)
        flags = _lib.F_INTERNAL_PIXELS | (_lib.F_DEVICE_OUTPUT if out_dev is not None else 0)
        if self.share_prefix if share_prefix is None else share_prefix:
            flags |= _lib.F_SHARE_PREFIX
        _lib.check(self.lib.vstar_vsm_score_batch(
            self.handle, B, None, None, ids.ctypes.data_as(ctypes.c_void_p), ids.shape[1],
            loc.ctypes.data_as(ctypes.c_void_p), vptr, nv, flags, out_ptr), self.handle)
        if out_dev is not None:
            return None
        return out if raw else self.unpack(out, nv)

    def preprocess_boxes(self, boxes_xyxy, slots=None) -> None:
        """Crop + pad + resize + normalise on the GPU into the engine's pixel buffers (consumed by a following call with
        internal pixels: score_grouped(internal_pixels=True))."""
        boxes = np.ascontiguousarray(np.asarray(boxes_xyxy, dtype=np.int32)).reshape(-1, 4)
        self._preprocess(boxes, slots)

    def preprocess_only(self, boxes_xyxy, slots=None):
        """(clip [B,3,I,I], owl [B,3,768,768]) float32 views of the device-side preprocessing result (tests)."""
        boxes = np.ascontiguousarray(np.asarray(boxes_xyxy, dtype=np.int32)).reshape(-1, 4)
        B = boxes.shape[0]
        self._preprocess(boxes, slots)
        I, O = self.cfg.clip_image_size, self.cfg.owl_image_size
        return (self.debug_read("clip_pixels", B * 3 * I * I).reshape(B, 3, I, I),
                self.debug_read("owl_pixels", B * 3 * O * O).reshape(B, 3, O, O))

    # ---- the search loop's one collective, inside the C-ABI (include/vstar_hip.h: vstar_comm_*, vstar_allgather_results) ----
    def comm_unique_id(self) -> bytes:
        buf = (ctypes.c_uint8 * 128)()
        _lib.check(self.lib.vstar_comm_unique_id(buf), self.handle)
        return bytes(buf)

    def comm_init(self, uid: bytes, world: int, rank: int) -> None:
        """Collective: every rank calls it with rank 0's id (vstar_amd.dist.engine_comm_init distributes it)."""
        assert len(uid) == 128
        buf = (ctypes.c_uint8 * 128).from_buffer_copy(uid)
        _lib.check(self.lib.vstar_comm_init(self.handle, buf, int(world), int(rank)), self.handle)
        self.comm_world, self.comm_rank = int(world), int(rank)

    def allgather_results(self, local_dev: torch.Tensor, sync: bool = True) -> torch.Tensor:
        """local_dev [n_local, RESULT_FLOATS] fp32 on this engine's device -> [world * n_local, RESULT_FLOATS] (rank-major), by
        ncclAllGather on the engine's stream."""
        assert local_dev.is_cuda and local_dev.dtype == torch.float32 and local_dev.is_contiguous() and local_dev.shape[1] == _lib.RESULT_FLOATS
        n = local_dev.shape[0]
        out = torch.empty((self.comm_world * n, _lib.RESULT_FLOATS), dtype=torch.float32, device=local_dev.device)
        _lib.check(self.lib.vstar_allgather_results(self.handle, ctypes.c_void_p(local_dev.data_ptr()), n, ctypes.c_void_p(out.data_ptr()),
                                                    0 if sync else _lib.F_NO_SYNC), self.handle)
        return out

    comm_world = 0
    comm_rank = 0

    @staticmethod
    def lib_result_floats() -> int:
        return _lib.RESULT_FLOATS

    @staticmethod
    def unpack(rec: np.ndarray, n_verify: int = 0) -> Dict[str, np.ndarray]:
        B = rec.shape[0]
        o0, o1, o2 = N_BOXES, N_BOXES * 5, N_BOXES * 5 + MASK_RES * MASK_RES
        return {
            "pred_logits": rec[:, :o0].reshape(B, N_BOXES, 1),
            "pred_boxes": rec[:, o0:o1].reshape(B, N_BOXES, 4),
            "low_res_masks": rec[:, o1:o2].reshape(B, 1, MASK_RES, MASK_RES),
            "tf_argmax": rec[:, o2:o2 + MAX_VERIFY].view(np.int32)[:, :n_verify].copy(),
        }

    def upsample_mask(self, low_res: np.ndarray, h: int, w: int, clamp: bool = True) -> np.ndarray:
        """F.interpolate(low_res.float(), (h, w), bilinear, align_corners=False) [+ clamp(min=0)] on the GPU."""
        src = np.ascontiguousarray(low_res, dtype=np.float32).reshape(MASK_RES, MASK_RES)
        out = np.empty((h, w), dtype=np.float32)
        _lib.check(self.lib.vstar_upsample_mask_ex(self.handle, src.ctypes.data_as(ctypes.c_void_p), h, w, 1 if clamp else 0,
                                                   out.ctypes.data_as(ctypes.c_void_p)), self.handle)
        return out

    def heatmap_stats(self, low_res: np.ndarray, h: int, w: int, rects_xywh=None) -> np.ndarray:
        """[min, max, sum, sum over each rect] of clamp(bilinear(low_res -> h x w), 0), computed on the GPU in fp64."""
        src = np.ascontiguousarray(low_res, dtype=np.float32).reshape(MASK_RES, MASK_RES)
        rects = np.zeros((0, 4), np.int32) if rects_xywh is None else np.ascontiguousarray(np.asarray(rects_xywh, np.int32)).reshape(-1, 4)
        out = np.zeros((3 + len(rects),), dtype=np.float64)
        _lib.check(self.lib.vstar_heatmap_stats(self.handle, src.ctypes.data_as(ctypes.c_void_p), h, w, len(rects),
                                                rects.ctypes.data_as(ctypes.c_void_p) if len(rects) else None,
                                                out.ctypes.data_as(ctypes.c_void_p)), self.handle)
        return out

    def heatmap_stats_batch(self, items) -> List[np.ndarray]:
        """items: [(low_res [192,192], h, w, rects_xywh or None), ...] -> heatmap_stats of each, in ONE engine call."""
        n = len(items)
        if n == 0:
            return []
        low = np.empty((n, MASK_RES, MASK_RES), np.float32)
        hw = np.empty((n, 2), np.int32)
        nr = np.zeros((n,), np.int32)
        rects = np.zeros((n, 8, 4), np.int32)
        for i, (m, h, w, r) in enumerate(items):
            low[i] = np.asarray(m, np.float32).reshape(MASK_RES, MASK_RES)
            hw[i] = (h, w)
            if r is not None and len(r):
                rr = np.asarray(r, np.int32).reshape(-1, 4)
                nr[i] = len(rr)
                rects[i, :len(rr)] = rr
        out = np.zeros((n, 11), np.float64)
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
        _lib.check(self.lib.vstar_heatmap_stats_batch(self.handle, n, p(low), p(hw), p(nr), p(rects), p(out)), self.handle)
        return [out[i, :3 + nr[i]].copy() for i in range(n)]

    def debug_read(self, name: str, count: int) -> np.ndarray:
        out = np.empty((count,), dtype=np.float32)
        n = self.lib.vstar_debug_read(self.handle, name.encode(), out.ctypes.data_as(ctypes.c_void_p), count)
        if n < 0:
            _lib.check(int(n), self.handle)
        return out[:n]

    def profile(self, on: bool) -> None:
        _lib.check(self.lib.vstar_profile_enable(self.handle, 1 if on else 0), self.handle)

    def profile_read(self):
        ms, n, fl = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
        _lib.check(self.lib.vstar_profile_read(self.handle, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl)), self.handle)
        return ms.value, n.value, fl.value

    def profile_read_fp8(self):
        """(ms, launches, flops) of the W8A8 launches inside the current GEMM profile (vstar_profile_read_fp8)."""
        ms, n, fl = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
        _lib.check(self.lib.vstar_profile_read_fp8(self.handle, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl)), self.handle)
        return ms.value, n.value, fl.value

    def w8a8_mx_active(self) -> int:
        """Which W8A8 activation scheme the last scoring step ran (csrc/mx.hpp; vstar_w8a8_mx_active): 0 = per-token scales (or bf16:
        fewer than 1024 rows / rows % 256 != 0), 1 = block-scaled inputs of o_proj / down_proj, 2 = the fully block-scaled chain
        (q|k|v and gate|up too, RMSNorms folded)."""
        return int(self.lib.vstar_w8a8_mx_active(self.handle))

    @property
    def stream(self) -> int:
        return int(self.lib.vstar_stream(self.handle) or 0)

    def close(self) -> None:
        if self.handle:
            self.lib.vstar_destroy(self.handle)
            self.handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def loc_positions(input_ids: np.ndarray, loc_token_idx: int, n_img_tokens: int) -> np.ndarray:
    """Spliced-sequence index of the hidden state that predicts [LOC]: idx([LOC]) - 1 + (P - 1)
    (the `255` shift of VSM.py:230-234,466-473 generalised to P-1)."""
    ids = np.asarray(input_ids)
    out = np.empty((ids.shape[0],), dtype=np.int32)
    for b in range(ids.shape[0]):
        w = np.where(ids[b] == loc_token_idx)[0]
        if w.size == 0:
            raise IndexError("no [LOC] token in the sequence (the reference fails on pred_mask[-1] here too)")
        out[b] = int(w[-1]) - 1 + (n_img_tokens - 1)
    return out


__all__ = ["VstarEngine", "loc_positions", "IMAGE_TOKEN_INDEX"]
