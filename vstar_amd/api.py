"""Model-level loading API of the reference, on the HIP engines — so that the reference's own wrapper classes
(`visual_search.py::VSM`, `vstar_bench_eval.py::VQA_LLM`) can keep their bodies and only swap the import:

    from vstar_amd.api import VSMForCausalLM            # VisualSearch/model/VSM.py:366-553 as used by visual_search.py:157-207
    from vstar_amd.api import load_pretrained_model     # LLaVA/llava/model/builder.py:26-151 as used by vstar_bench_eval.py:38-47

Both return thin facades over the engines (no torch modules, no CPU compute path): the objects expose exactly the members those
wrapper classes touch.  The batched fast paths remain `vstar_amd.vsm.VSM` / `vstar_amd.vqa.VQA_LLM`; these facades are the
literal batch-1 call shapes of the reference.

Out of scope (raise): 8-bit / 4-bit loading (bitsandbytes), LoRA `model_base` merging, the MPT variant, sampling / beams.
"""
from __future__ import annotations

import os
from types import SimpleNamespace
from typing import List, Optional

import numpy as np
import torch

from .config import VSMConfig
from .engine import VstarEngine
from .weights import load_checkpoint_dir, random_state_dict


def _device_index(device) -> int:
    if isinstance(device, int):
        return device
    s = str(device)
    return int(s.split(":")[1]) if ":" in s else 0


# =====================================================================================================================
# VSMForCausalLM
# =====================================================================================================================
class _ClipProcessorHandle:
    """`vision_tower.image_processor` (visual_search.py:163): the CLIP preprocessing of vstar_amd.preprocess behind the
    CLIPImageProcessor call shape the reference uses (`.preprocess(img, return_tensors="pt")["pixel_values"][0]`)."""
    from .preprocess import CLIP_MEAN as image_mean, CLIP_STD as image_std  # noqa: N813

    def __init__(self, size: int):
        self.size = size
        self.crop_size = {"height": size, "width": size}

    def preprocess(self, image, return_tensors: str = "pt"):
        from .preprocess import clip_preprocess
        # the reference passes expand2square(image) here; clip_preprocess applies the same (idempotent) padding first
        return {"pixel_values": [torch.from_numpy(clip_preprocess(image, self.size))]}


class _VisionTowerHandle:
    def __init__(self, size: int):
        self.image_processor = _ClipProcessorHandle(size)
        self.is_loaded = True

    def cuda(self):
        return self

    def to(self, *a, **k):
        return self


class VSMForCausalLM:
    """`VSMForCausalLM.from_pretrained(version, low_cpu_mem_usage=True, vision_tower=..., loc_token_idx=..., torch_dtype=bf16,
    device_map='cuda', is_eval=True)` then `.get_model().initialize_vision_modules(cfg)`, `.get_model().get_vision_tower()`,
    `.eval()`, `.config.vision_tower`, and

        output_ids, pred_masks, det_result = model.inference(images_clip, images, input_ids, resize_list, original_size_list,
                                                             max_new_tokens=100, tokenizer=tok, mode=mode)

    with the reference's return conventions (VSM.py:438-553): mode 'vqa' -> (output_ids [1, L+n], None, None); 'segmentation'
    -> (None, [masks [n_loc, h, w] fp32, NOT clamped], None); 'detection' -> (None, masks, {'pred_logits' [n_loc,2304,1],
    'pred_boxes' [n_loc,2304,4]} in bf16).  n_loc = number of [LOC] tokens greedy decoding emits (0 -> empty tensors: the
    caller's pred_mask[-1] raises IndexError exactly like the reference).

    Mechanism: the reference's generate(use_cache=False) re-runs CLIP + a full prefill per new token; here the answer is
    decoded once with the KV cache (vstar_vsm_generate: same arg-max tokens) and the crop is scored by ONE prefill with the
    generated answer teacher-forced — under causal attention the hidden state in front of each [LOC] is the same."""

    def __init__(self, engine: VstarEngine, loc_token_idx: int, vision_tower_name: str = "openai/clip-vit-large-patch14",
                 eos_token_id: int = 2):
        self.engine, self.cfg = engine, engine.cfg
        self.loc_token_idx = int(loc_token_idx)
        self.eos_token_id = eos_token_id
        self.config = SimpleNamespace(vision_tower=vision_tower_name, mm_vision_tower=vision_tower_name,
                                      hidden_size=self.cfg.llm_hidden, vocab_size=self.cfg.llm_vocab)
        self._tower = _VisionTowerHandle(self.cfg.clip_image_size)

    @classmethod
    def from_pretrained(cls, version, low_cpu_mem_usage: bool = True, vision_tower: Optional[str] = None,
                        loc_token_idx: Optional[int] = None, torch_dtype=None, device_map="cuda", is_eval: bool = True, *,
                        cfg: Optional[VSMConfig] = None, state_dict=None, synthetic_seed: Optional[int] = None, device: int = 0,
                        **unused):
        """`version`: LOCAL HF checkpoint directory of craigwu/seal_vsm_7b; `vision_tower`: LOCAL openai/clip-vit-large-patch14
        directory (no hub access here).  Offline: pass `state_dict` (engine key space) or `synthetic_seed`."""
        if loc_token_idx is None:
            raise ValueError("loc_token_idx is required (visual_search.py:156-158)")
        if torch_dtype not in (None, torch.bfloat16):
            raise NotImplementedError("the VSM engine computes in bfloat16 (the reference's torch_dtype, visual_search.py:145)")
        cfg = cfg or VSMConfig.seal_7b(224)
        if state_dict is None:
            if version is not None and os.path.isdir(str(version)):
                if vision_tower is None or not os.path.isdir(str(vision_tower)):
                    raise FileNotFoundError(f"vision_tower {vision_tower!r} must be a local openai/clip-vit-large-patch14 directory")
                state_dict = load_checkpoint_dir(str(version), str(vision_tower))
            elif synthetic_seed is not None:
                state_dict = random_state_dict(cfg, seed=synthetic_seed, dtype=torch.bfloat16, share_layers=True)
            else:
                raise FileNotFoundError(f"checkpoint directory {version!r} not found and neither state_dict nor synthetic_seed given")
        eng = VstarEngine(cfg, _device_index(device if device_map in ("cuda", "auto", None) else device_map))
        eng.load_state_dict(state_dict)
        return cls(eng, loc_token_idx, vision_tower_name=str(vision_tower or "openai/clip-vit-large-patch14"))

    # ---- the members visual_search.py:160-166 touches ----
    def get_model(self):
        return self

    def initialize_vision_modules(self, config=None):
        return None                      # the CLIP tower was packed with the other weights

    def get_vision_tower(self):
        return self._tower

    def eval(self):
        return self

    # ---- VSM.py:438-553 ----
    @torch.inference_mode()
    def inference(self, images_clip, images, input_ids, resize_list=None, original_size_list=None, max_new_tokens: int = 32,
                  tokenizer=None, mode: str = "vqa"):
        assert mode in ("vqa", "segmentation", "detection")
        ids_t = torch.as_tensor(input_ids)
        if ids_t.dim() != 2 or ids_t.shape[0] != 1:
            raise ValueError("inference() takes one crop per call, like the reference (input_ids [1, L])")
        prompt = [int(v) for v in ids_t[0].tolist()]
        clip = torch.as_tensor(images_clip).to(torch.bfloat16).cpu().reshape(1, 3, self.cfg.clip_image_size, self.cfg.clip_image_size)
        room = self.cfg.max_text_len - len(prompt)
        eos = getattr(tokenizer, "eos_token_id", None) or self.eos_token_id
        gen: List[int] = self.engine.generate(clip, prompt, min(max_new_tokens, room), eos) if room > 0 else []
        output_ids = torch.tensor([prompt + gen], dtype=torch.long)
        if mode == "vqa":
            return output_ids, None, None
        h, w = (int(v) for v in original_size_list[0])
        locs = [k for k, t in enumerate(gen) if t == self.loc_token_idx]
        if not locs:
            empty = torch.zeros((0, h, w), dtype=torch.float32)
            det = {"pred_logits": torch.zeros((0, 2304, 1), dtype=torch.bfloat16), "pred_boxes": torch.zeros((0, 2304, 4), dtype=torch.bfloat16)}
            return None, [empty], (None if mode == "segmentation" else det)
        owl = torch.as_tensor(images).to(torch.bfloat16).cpu().reshape(1, 3, self.cfg.owl_image_size, self.cfg.owl_image_size)
        P = self.cfg.n_img_tokens
        ids = np.asarray(prompt + gen[:locs[-1] + 1], np.int32)
        pos = [len(prompt) + k - 1 + (P - 1) for k in locs]
        logits, boxes, masks = [], [], []
        mb = self.cfg.max_batch
        for s0 in range(0, len(pos), mb):
            chunk = pos[s0:s0 + mb]
            B = len(chunk)
            res = self.engine.score_batch(clip.repeat(B, 1, 1, 1), owl.repeat(B, 1, 1, 1), np.tile(ids[None], (B, 1)),
                                          np.asarray(chunk, np.int32))
            logits.append(res["pred_logits"])
            boxes.append(res["pred_boxes"])
            masks += [self.engine.upsample_mask(res["low_res_masks"][b, 0], h, w, clamp=False) for b in range(B)]
        pred_masks = [torch.from_numpy(np.stack(masks))]
        if mode == "segmentation":
            return None, pred_masks, None
        det = {"pred_logits": torch.from_numpy(np.concatenate(logits)).to(torch.bfloat16),
               "pred_boxes": torch.from_numpy(np.concatenate(boxes)).to(torch.bfloat16)}
        return None, pred_masks, det


# =====================================================================================================================
# load_pretrained_model (SEAL VQA-LLM)
# =====================================================================================================================
class _PastKeyValues:
    """Opaque stand-in for HF's past_key_values: the KV-cache slot of the engine that holds the positions [0, length)."""

    def __init__(self, slot: int, length: int, forked: bool):
        self.slot, self.length, self.forked = slot, length, forked


class LlavaSearchModel:
    """The slice of LlavaSearchLlamaForCausalLM that vstar_bench_eval.py:78-165 touches: `generate(...)`, `model(...)` with
    `.logits` / `.past_key_values`, `.config.vocab_size`.  fp16, KV cache in HBM; a continuation (`past_key_values=`) FORKS the
    question's slot (no copy, no re-prefill), which is what the option scoring of the evaluation needs."""

    MAX_LOGIT_ROWS = 256          # logits rows one engine call returns (vstar_vqa_forward's n_want limit)

    def __init__(self, llm):
        self._llm = llm
        self.engine, self.cfg = llm.engine, llm.cfg
        self.config = SimpleNamespace(vocab_size=self.cfg.llm_vocab, hidden_size=self.cfg.llm_hidden, mm_use_im_start_end=False,
                                      mm_use_im_patch_token=True)
        self._next_fork = 1

    def get_vision_tower(self):
        return _VisionTowerHandle(self.cfg.clip_image_size)

    def eval(self):
        return self

    def _rows(self, input_ids, images, object_features, images_long, objects_long):
        ids = [int(v) for v in torch.as_tensor(input_ids)[0].tolist()]
        pix = [torch.as_tensor(images).reshape(-1, 3, self.cfg.clip_image_size, self.cfg.clip_image_size)]
        n_img = pix[0].shape[0]
        n_obj = 0
        if object_features is not None and len(object_features) > 0:
            pix.append(torch.as_tensor(object_features).reshape(-1, 3, self.cfg.clip_image_size, self.cfg.clip_image_size))
            n_obj = pix[1].shape[0]
        if n_img + n_obj > self.cfg.max_images:
            raise ValueError("more images / object crops than feature slots (VQAConfig.max_images)")
        self.engine.encode_images(torch.cat(pix, 0).float(), 0)
        img_slots, obj_slots = list(range(n_img)), list(range(n_img, n_img + n_obj))
        return ids, self.engine.expand_ids(ids, img_slots, obj_slots, images_long, objects_long)

    @torch.inference_mode()
    def __call__(self, input_ids=None, use_cache: bool = True, images=None, object_features=None, images_long=None,
                 objects_long=None, past_key_values: Optional[_PastKeyValues] = None, attention_mask=None, **unused):
        from .vqa_engine import Seq
        if past_key_values is None:
            _, rows = self._rows(input_ids, images, object_features, images_long, objects_long)
            seq, past = Seq(rows, kv_slot=0), _PastKeyValues(0, len(rows), forked=False)
            self._next_fork = 1
        else:
            if past_key_values.forked:
                raise NotImplementedError("continuing a forked continuation (two levels) is not used by the evaluation")
            rows = [int(v) for v in torch.as_tensor(input_ids)[0].tolist()]
            slot = self._next_fork
            self._next_fork = 1 + (self._next_fork % (self.cfg.max_slots - 1))
            seq = Seq(rows, kv_slot=slot, past_len=past_key_values.length, prefix_slot=past_key_values.slot)
            past = _PastKeyValues(slot, past_key_values.length + len(rows), forked=True)
        # lm_head runs only on requested rows (at most MAX_LOGIT_ROWS per call): `.logits` has the full [1, T, vocab] shape the
        # reference returns, with the rows in front of the last MAX_LOGIT_ROWS left NaN (the evaluation reads logits[:, -1:] of
        # the question and every row of the short option continuations)
        T = len(rows)
        first = max(0, T - self.MAX_LOGIT_ROWS)
        got, _ = self.engine.forward([seq], [(0, t) for t in range(first, T)])
        logits = torch.full((1, T, self.cfg.llm_vocab), float("nan"), dtype=torch.float16)
        logits[0, first:] = torch.from_numpy(got)
        return SimpleNamespace(logits=logits, past_key_values=past if use_cache else None)

    @torch.inference_mode()
    def generate(self, input_ids, images=None, object_features=None, images_long=None, objects_long=None, do_sample: bool = False,
                 num_beams: int = 1, temperature: float = 0, top_p=None, max_new_tokens: int = 200, use_cache: bool = True,
                 stopping_criteria=None, **unused):
        from .vqa_engine import Seq
        if do_sample or num_beams != 1:
            raise NotImplementedError("the evaluation decodes greedily (temperature 0, one beam)")
        ids, rows = self._rows(input_ids, images, object_features, images_long, objects_long)
        new = self._llm.greedy_decode([Seq(rows, kv_slot=0)], [len(rows)], max_new_tokens)[0]
        return torch.tensor([ids + new], dtype=torch.long)


def load_pretrained_model(model_path, model_base=None, model_name: str = "", load_8bit: bool = False, load_4bit: bool = False,
                          device_map="auto", device="cuda", *, cfg=None, state_dict=None, tokenizer=None, vision_tower=None):
    """Same signature and return tuple as LLaVA/llava/model/builder.py:26-151: (tokenizer, model, image_processor, context_len).
    `model_path`: LOCAL checkpoint directory of craigwu/seal_vqa_7b (offline: `cfg` + `state_dict` [+ `tokenizer`])."""
    if load_8bit or load_4bit:
        raise NotImplementedError("bitsandbytes 8-bit / 4-bit loading is out of scope: the engine runs fp16 (builder.py:43)")
    if model_base is not None:
        raise NotImplementedError("LoRA / model_base merging is out of scope (the evaluation passes model_base=None)")
    if "mpt" in model_name.lower():
        raise NotImplementedError("the MPT variant is out of scope")
    from .vqa import VQA_LLM
    llm = VQA_LLM(SimpleNamespace(vqa_model_path=model_path, conv_type="v1", vision_tower=vision_tower), cfg=cfg,
                  state_dict=state_dict, tokenizer=tokenizer, device=_device_index(device))
    return llm.tokenizer, LlavaSearchModel(llm), llm.image_processor, llm.context_len


__all__ = ["VSMForCausalLM", "load_pretrained_model", "LlavaSearchModel"]
