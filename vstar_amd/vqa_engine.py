"""Python face of the VQA-LLM HIP engine (include/vstar_vqa.h): weight hand-over, image/object feature encoding into the
device-resident feature table, and the KV-cached forward over ragged rows of many sequences.  All arithmetic happens in
libvstar_hip.so (fp16 instantiation); numpy/torch only hold host buffers."""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from .config import IMAGE_TOKEN_INDEX, OBJECT_TOKEN_INDEX, PAD_ROW, VQAConfig
from .weights import vqa_state_dict_spec

_DT = {torch.float32: _lib.F32, torch.float16: _lib.F16, torch.bfloat16: _lib.BF16}


@dataclass
class Seq:
    """New rows of one sequence for `VqaEngine.forward`."""
    rows: Sequence[int]          # >= 0 vocabulary ids; < 0 feature rows as produced by VqaEngine.feature_rows()
    kv_slot: int                 # cache slot receiving the new rows' K/V
    past_len: int = 0            # cached positions in front of the new rows
    prefix_slot: Optional[int] = None   # slot holding [0, past_len); None = kv_slot


class VqaEngine:
    def __init__(self, cfg: VQAConfig, device: int = 0):
        self.cfg = cfg
        self.lib = _lib.load()
        self.handle = ctypes.c_void_p()
        c = cfg.to_c()
        _lib.check_vqa(self.lib.vstar_vqa_create(ctypes.byref(c), device, ctypes.byref(self.handle)))
        self.device = device
        self.finalized = False

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.vstar_vqa_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    # ---- weights (replaces load_pretrained_model, LLaVA/llava/model/builder.py:26-151) ----
    def load_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        for k in vqa_state_dict_spec(self.cfg):
            if k.startswith("clip.vision_model.encoder.layers.") and \
                    int(k.split(".")[4]) >= self.cfg.clip_layers + 1 + self.cfg.clip_select_layer:
                continue
            if k.startswith("clip.vision_model.post_layernorm"):
                continue
            if k not in sd:
                raise KeyError(f"checkpoint tensor missing: {k}")
            t = sd[k].detach().cpu().contiguous()
            if t.dtype not in _DT:
                t = t.float()
            shape = (ctypes.c_int64 * max(t.dim(), 1))(*t.shape)
            _lib.check_vqa(self.lib.vstar_vqa_load_tensor(self.handle, k.encode(), ctypes.c_void_p(t.data_ptr()), _DT[t.dtype],
                                                          t.dim(), shape), self.handle)
        _lib.check_vqa(self.lib.vstar_vqa_finalize_weights(self.handle), self.handle)
        self.finalized = True

    # ---- encode_images / project_features (llava_search_arch.py:84-94) ----
    def encode_images(self, pixels, first_slot: int = 0) -> None:
        """pixels [n,3,I,I] (CLIPImageProcessor output); fills feature slots first_slot .. first_slot+n-1."""
        t = torch.as_tensor(pixels).to(torch.float16).contiguous().cpu()
        I = self.cfg.clip_image_size
        assert t.dim() == 4 and tuple(t.shape[1:]) == (3, I, I), t.shape
        _lib.check_vqa(self.lib.vstar_vqa_encode_images(self.handle, t.shape[0], ctypes.c_void_p(t.data_ptr()), first_slot),
                       self.handle)

    def feature_rows(self, slot: int, long: bool) -> List[int]:
        """Row sources that splice the long (P rows) or short (pcv_latents rows) features of a slot."""
        P, L = self.cfg.n_img_tokens, self.cfg.pcv_latents
        base = slot * (P + L)
        idx = range(base, base + P) if long else range(base + P, base + P + L)
        return [-(1 + i) for i in idx]

    def expand_ids(self, ids: Sequence[int], image_slots: Sequence[int], object_slots: Sequence[int],
                   images_long: Optional[Sequence[bool]], objects_long: Optional[Sequence[bool]]) -> List[int]:
        """<image> (-200) / <object> (-300) placeholders -> feature rows, with the long/short selection of
        prepare_inputs_labels_for_multimodal (llava_search_arch.py:136-140,175-179)."""
        out: List[int] = []
        ii = io = 0
        for t in ids:
            if t == IMAGE_TOKEN_INDEX:
                out += self.feature_rows(image_slots[ii], images_long is None or bool(images_long[ii]))
                ii += 1
            elif t == OBJECT_TOKEN_INDEX:
                out += self.feature_rows(object_slots[io], not (objects_long is None or not objects_long[io]))
                io += 1
            else:
                out.append(int(t))
        return out

    # ---- LlavaSearchLlamaForCausalLM.forward over new rows (llava_search_llama.py:56-113) ----
    def forward(self, seqs: Sequence[Seq], want: Sequence[Tuple[int, int]], logits: bool = True):
        """want: (sequence index, row index inside that sequence's new rows; negative counts from the end).
        Returns (logits float16 [n_want, vocab] or None, argmax int32 [n_want])."""
        n = len(seqs)
        row_off = np.zeros(n + 1, np.int32)
        for i, s in enumerate(seqs):
            row_off[i + 1] = row_off[i] + len(s.rows)
        src = np.concatenate([np.asarray(s.rows, np.int64) for s in seqs]).astype(np.int32)
        kv = np.asarray([s.kv_slot for s in seqs], np.int32)
        pre = np.asarray([s.kv_slot if s.prefix_slot is None else s.prefix_slot for s in seqs], np.int32)
        past = np.asarray([s.past_len for s in seqs], np.int32)
        w = np.asarray([row_off[i] + (r if r >= 0 else len(seqs[i].rows) + r) for i, r in want], np.int32)
        nw = len(w)
        out_logits = np.empty((nw, self.cfg.llm_vocab), np.float16) if (logits and nw) else None
        out_arg = np.empty((max(nw, 1),), np.int32)
        P = lambda a: ctypes.c_void_p(a.ctypes.data) if a is not None and a.size else None
        _lib.check_vqa(self.lib.vstar_vqa_forward(self.handle, n, P(row_off), P(src), P(kv), P(pre), P(past), nw, P(w),
                                                  P(out_logits), P(out_arg)), self.handle)
        return out_logits, out_arg[:nw]

    def last_forward_ms(self) -> float:
        return float(self.lib.vstar_vqa_last_forward_ms(self.handle))

    def debug_read(self, name: str, count: int) -> np.ndarray:
        out = np.empty((count,), np.float32)
        n = self.lib.vstar_vqa_debug_read(self.handle, name.encode(), ctypes.c_void_p(out.ctypes.data), count)
        if n < 0:
            _lib.check_vqa(int(n), self.handle)
        return out[:n]

    def features(self, slot: int) -> Tuple[np.ndarray, np.ndarray]:
        """(long [P,H], short [L,H]) of one feature slot, as float32 (parity tests)."""
        P, L, H = self.cfg.n_img_tokens, self.cfg.pcv_latents, self.cfg.llm_hidden
        all_ = self.debug_read("features", (slot + 1) * (P + L) * H).reshape(slot + 1, P + L, H)
        return all_[slot, :P], all_[slot, P:]
