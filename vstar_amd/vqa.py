"""`VQA_LLM` — drop-in for the reference class of the same name (vstar_bench_eval.py:38-165) on the HIP engine.

Same constructor role, same methods and return conventions:
    get_patch / get_object_crop ............ vstar_bench_eval.py:49-77
    free_form_inference(image, question, ...) -> str ................ :78-113   (temperature 0: greedy, KV cache)
    multiple_choices_inference(image, question, options, ...) -> int  :115-165  (shared-prefix option scoring)
plus a batched form (`free_form_batch`) that decodes many samples in one engine call per step — the reference runs batch 1; on an MI355X a decode step is bound by the 13.5 GB weight sweep, so sequences are
advanced together.

MI355X-first differences that do not change results: the question prefix of the multiple-choice scoring is prefilled
once and every option forks its KV slot (no copy, no re-prefill); image/object features stay in HBM and are spliced by
row index; only the logits rows that are needed come back to the host.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
from PIL import Image

from .config import IMAGE_TOKEN_INDEX, OBJECT_TOKEN_INDEX, VQAConfig
from .preprocess import CLIP_MEAN, CLIP_STD, SyntheticTokenizer, _normalise
from .vqa_engine import Seq, VqaEngine

DEFAULT_IMAGE_TOKEN = "<image>"
DEFAULT_OBJECT_TOKEN = "<object>"

# conv_templates["v1"] = conv_vicuna_v1 (LLaVA/llava/conversation.py:252-262), SeparatorStyle.TWO (:51-60)
V1_SYSTEM = ("A chat between a curious user and an artificial intelligence assistant. "
             "The assistant gives helpful, detailed, and polite answers to the user's questions.")
V1_ROLES = ("USER", "ASSISTANT")
V1_SEP, V1_SEP2 = " ", "</s>"


def v1_prompt(user: str, answer: Optional[str] = None) -> str:
    ret = V1_SYSTEM + V1_SEP + V1_ROLES[0] + ": " + user + V1_SEP
    return ret + V1_ROLES[1] + (": " + answer + V1_SEP2 if answer else ":")


def tokenizer_image_object_token(prompt: str, tokenizer) -> List[int]:
    """Tokenises the text between the <image> / <object> markers separately and joins the pieces with the -200 / -300
    placeholders, keeping one BOS (LLaVA/llava/mm_utils.py:64-88)."""
    pieces: List[str] = []
    marks: List[int] = []
    for gi, group in enumerate(prompt.split(DEFAULT_IMAGE_TOKEN)):
        for oi, piece in enumerate(group.split(DEFAULT_OBJECT_TOKEN)):
            if pieces:
                # the reference's separator list is [image] + [object] * (n-1): the first boundary is the image one
                marks.append(IMAGE_TOKEN_INDEX if len(marks) == 0 else OBJECT_TOKEN_INDEX)
            pieces.append(piece)
    toks = [tokenizer(p).input_ids for p in pieces]
    bos = getattr(tokenizer, "bos_token_id", None)
    has_bos = bool(toks and toks[0] and toks[0][0] == bos)
    ids: List[int] = [bos] if has_bos else []
    for i, t in enumerate(toks):
        if i > 0:
            ids.append(marks[i - 1])
        ids.extend(t[1:] if has_bos else t)
    return ids


class _ImageProcessor:
    """The slice of CLIPImageProcessor the evaluation touches (vstar_bench_eval.py:75-76,88,125,191)."""
    image_mean = list(CLIP_MEAN)
    image_std = list(CLIP_STD)

    def __init__(self, size: int = 224):
        self.crop_size = {"height": size, "width": size}
        self.size = size

    def preprocess(self, image: Image.Image, return_tensors: str = "pt"):
        img = image.convert("RGB")
        w, h = img.size
        short, long_ = (w, h) if w <= h else (h, w)
        ns, nl = self.size, int(self.size * long_ / short)
        nw, nh = (ns, nl) if w <= h else (nl, ns)
        img = img.resize((nw, nh), resample=Image.BICUBIC)
        left, top = (nw - self.size) // 2, (nh - self.size) // 2
        img = img.crop((left, top, left + self.size, top + self.size))
        return {"pixel_values": [torch.from_numpy(_normalise(np.asarray(img)))]}


class VQA_LLM:
    def __init__(self, args=None, cfg: Optional[VQAConfig] = None, state_dict: Optional[Dict[str, torch.Tensor]] = None,
                 tokenizer=None, engine: Optional[VqaEngine] = None, device: int = 0):
        """args: the reference's namespace (vqa_model_path, conv_type).  With a local checkpoint directory the weights and
        tokenizer are read from it; offline (this environment) pass `state_dict` (+ optionally `tokenizer`)."""
        import os
        from .weights import load_vqa_checkpoint_dir, vqa_config_from_dir
        path = getattr(args, "vqa_model_path", None) if args is not None else None
        real = engine is None and state_dict is None and path is not None and os.path.isdir(str(path))
        if real:                      # load_pretrained_model(model_path, None, name) (builder.py:26-151), local files only
            cfg = cfg or vqa_config_from_dir(path)
            state_dict = load_vqa_checkpoint_dir(path, getattr(args, "vision_tower", None))
            if tokenizer is None:
                from transformers import AutoTokenizer
                tokenizer = AutoTokenizer.from_pretrained(path, use_fast=False)
                tokenizer.add_tokens(["<im_patch>"], special_tokens=True)      # mm_use_im_patch_token default (builder.py:131-133)
        self.cfg = cfg or (engine.cfg if engine is not None else VQAConfig.seal_7b())
        self.conv_type = getattr(args, "conv_type", "v1") if args is not None else "v1"
        if self.conv_type != "v1":
            raise ValueError("only the 'v1' conversation template of the reference evaluation is implemented")
        self.tokenizer = tokenizer or SyntheticTokenizer(self.cfg.llm_vocab)
        self.image_processor = _ImageProcessor(self.cfg.clip_image_size)
        self.context_len = 2048
        if engine is None:
            if state_dict is None:
                raise FileNotFoundError(f"VQA-LLM checkpoint directory {path!r} not found and no state_dict given "
                                        "(there is no hub access here; the engine has no CPU fallback)")
            engine = VqaEngine(self.cfg, device)
            engine.load_state_dict(state_dict)
        self.engine = engine
        self.model = SimpleNamespace(config=SimpleNamespace(vocab_size=self.cfg.llm_vocab))
        self.eos_token_id = getattr(self.tokenizer, "eos_token_id", 2)

    # ---- vstar_bench_eval.py:49-77 ----
    def get_patch(self, bbox, image_width, image_height, patch_size=224, patch_scale=None):
        object_width, object_height = int(np.ceil(bbox[2])), int(np.ceil(bbox[3]))
        cx, cy = int(bbox[0] + bbox[2] / 2), int(bbox[1] + bbox[3] / 2)
        if patch_scale is None:
            pw, ph = max(object_width, patch_size), max(object_height, patch_size)
        else:
            pw, ph = int(object_width * patch_scale), int(object_height * patch_scale)
        left = max(0, cx - pw // 2)
        right = min(left + pw, image_width)
        top = max(0, cy - ph // 2)
        bottom = min(top + ph, image_height)
        return [left, top, right, bottom]

    def get_object_crop(self, image, bbox, patch_scale):
        box = self.get_patch(bbox, image.width, image.height, patch_scale=patch_scale)
        crop = image.crop((box[0], box[1], box[2], box[3]))
        crop = crop.resize((self.image_processor.crop_size["width"], self.image_processor.crop_size["height"]))
        return self.image_processor.preprocess(crop, return_tensors="pt")["pixel_values"][0]

    # ---- shared plumbing ----
    def _encode(self, image, object_crops, first_slot: int):
        """Features of one sample into consecutive feature slots: image first, then its object crops."""
        pix = [self.image_processor.preprocess(image, return_tensors="pt")["pixel_values"][0]]
        n_obj = 0
        if object_crops is not None and len(object_crops) > 0:
            pix += [torch.as_tensor(c) for c in object_crops]
            n_obj = len(object_crops)
        if first_slot + 1 + n_obj > self.cfg.max_images:
            raise ValueError(f"{1 + n_obj} images/object crops exceed the engine's feature table (max_images="
                             f"{self.cfg.max_images}); build the engine with a larger VQAConfig.max_images")
        self.engine.encode_images(torch.stack(pix, 0), first_slot)
        return [first_slot], list(range(first_slot + 1, first_slot + 1 + n_obj))

    def _question_rows(self, question: str, img_slots, obj_slots, images_long, objects_long, answer: Optional[str] = None):
        ids = tokenizer_image_object_token(v1_prompt(DEFAULT_IMAGE_TOKEN + "\n" + question, answer), self.tokenizer)
        return ids, self.engine.expand_ids(ids, img_slots, obj_slots, images_long, objects_long)

    # ---- free-form answer (vstar_bench_eval.py:78-113) ----
    def free_form_inference(self, image, question, temperature=0, top_p=None, num_beams=1, max_new_tokens=200,
                            object_crops=None, images_long=None, objects_long=None) -> str:
        if temperature != 0 or num_beams != 1:
            raise NotImplementedError("the evaluation decodes greedily (temperature 0, one beam)")
        return self.free_form_batch([dict(image=image, question=question, object_crops=object_crops, images_long=images_long,
                                          objects_long=objects_long)], max_new_tokens)[0]

    def free_form_batch(self, samples: Sequence[dict], max_new_tokens: int = 200) -> List[str]:
        """Greedy decode of several samples at once: one prefill call, then one engine call per generated position."""
        cfg, eng = self.cfg, self.engine
        n = len(samples)
        if n > cfg.max_slots:
            raise ValueError("more samples than KV slots")
        seqs, lens = [], []
        fslot = 0
        for i, s in enumerate(samples):
            crops = s.get("object_crops")
            img_slots, obj_slots = self._encode(s["image"], crops, fslot)
            fslot += 1 + len(obj_slots)
            _, rows = self._question_rows(s["question"], img_slots, obj_slots, s.get("images_long"), s.get("objects_long"))
            seqs.append(Seq(rows, kv_slot=i))
            lens.append(len(rows))
        self.generated_ids = self.greedy_decode(seqs, lens, max_new_tokens)
        texts = []
        for ids in self.generated_ids:
            out = self.tokenizer.batch_decode([ids], skip_special_tokens=True)[0].strip()
            if out.endswith(V1_SEP2):
                out = out[:-len(V1_SEP2)]
            texts.append(out.strip())
        return texts

    def greedy_decode(self, seqs: Sequence[Seq], lens: Sequence[int], max_new_tokens: int) -> List[List[int]]:
        """model.generate(do_sample=False, use_cache=True) for every sequence; a sequence stops at EOS (the reference's
        keyword criterion stops on '</s>', the decoded EOS) or when the context is full."""
        eng, cfg = self.engine, self.cfg
        n = len(seqs)
        _, nxt = eng.forward(seqs, [(i, -1) for i in range(n)], logits=False)
        out: List[List[int]] = [[] for _ in range(n)]
        pos = list(lens)
        live = list(range(n))
        cur = {i: int(nxt[i]) for i in range(n)}
        for _ in range(max_new_tokens):
            still = []
            for i in live:
                out[i].append(cur[i])
                if cur[i] != self.eos_token_id and pos[i] + 1 < cfg.max_ctx:
                    still.append(i)
            live = still
            if not live or len(out[live[0]]) >= max_new_tokens:
                break
            step = [Seq([cur[i]], kv_slot=seqs[i].kv_slot, past_len=pos[i]) for i in live]
            _, nxt = eng.forward(step, [(j, 0) for j in range(len(live))], logits=False)
            for j, i in enumerate(live):
                cur[i] = int(nxt[j])
                pos[i] += 1
        return out

    # ---- multiple choice (vstar_bench_eval.py:115-165) ----
    def multiple_choices_inference(self, image, question, options, object_crops=None, images_long=None,
                                   objects_long=None) -> int:
        losses = self.option_losses(image, question, options, object_crops, images_long, objects_long)
        return int(torch.stack(losses).argmin().item())

    def option_losses(self, image, question, options, object_crops=None, images_long=None, objects_long=None):
        eng, cfg = self.engine, self.cfg
        if 1 + len(options) > cfg.max_slots:
            raise ValueError("more options than KV slots")
        img_slots, obj_slots = self._encode(image, object_crops, 0)
        q_ids, q_rows = self._question_rows(question, img_slots, obj_slots, images_long, objects_long)
        q_logits, _ = eng.forward([Seq(q_rows, kv_slot=0)], [(0, -1)])
        P = len(q_rows)
        seqs, want, opt_ids = [], [], []
        for j, option in enumerate(options):
            full_ids, _ = self._question_rows(question, img_slots, obj_slots, images_long, objects_long, answer=option)
            ids = full_ids[len(q_ids):]                                    # option_answer_input_ids (:145-146)
            opt_ids.append(ids)
            seqs.append(Seq(ids, kv_slot=1 + j, past_len=P, prefix_slot=0))
            want += [(j, t) for t in range(len(ids) - 1)]
        o_logits, _ = eng.forward(seqs, want) if want else (np.zeros((0, cfg.llm_vocab), np.float16), None)
        losses, k = [], 0
        for ids in opt_ids:
            rows = [torch.from_numpy(q_logits[0:1])]
            if len(ids) > 1:
                rows.append(torch.from_numpy(o_logits[k:k + len(ids) - 1]))
            k += len(ids) - 1
            lg = torch.cat(rows, 0)                                        # cat(question_logits[-1:], option_logits[:-1])
            # CrossEntropyLoss on the fp16 logits (:156-159): fp32 log-softmax internally, fp16 result
            loss = torch.nn.functional.cross_entropy(lg.float(), torch.tensor(ids, dtype=torch.long)).to(torch.float16)
            losses.append(loss)
        return losses
