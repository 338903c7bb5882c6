"""Host-side image preprocessing and prompt building of `VSM.inference` (visual_search.py:174-196).

PIL-exact restatements (the reference calls HF `CLIPImageProcessor` / `OwlViTProcessor`, both PIL bicubic on uint8):
  * expand2square — TOP-LEFT paste on a CLIP-mean canvas (VisualSearch/utils/utils.py:28-39; note: the VQA-LLM side
    centres the paste, vstar_bench_eval.py:25-36 — not this path)
  * CLIP processor — resize shortest edge to I (bicubic), centre-crop IxI, /255, normalise with the CLIP mean/std
  * OWL-ViT processor — resize to 768x768 (bicubic, aspect NOT preserved), /255, same normalisation
tests/test_host.py checks both against the installed HF processors.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np
from PIL import Image

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)

IMAGE_TOKEN_INDEX = -200
DEFAULT_IMAGE_TOKEN = "<image>"
DEFAULT_IM_START_TOKEN = "<im_start>"
DEFAULT_IM_END_TOKEN = "<im_end>"

# conv_templates["llava_v1"] (VisualSearch/model/llava/conversation.py:355-365), SeparatorStyle.TWO (:53-62)
LLAVA_V1_SYSTEM = ("A chat between a curious human and an artificial intelligence assistant. "
                   "The assistant gives helpful, detailed, and polite answers to the human's questions.")
LLAVA_V1_ROLES = ("USER", "ASSISTANT")
LLAVA_V1_SEP, LLAVA_V1_SEP2 = " ", "</s>"

LOCATE_QUESTION = "Please locate the {} in this image."                        # visual_search.py:396
CUE_QUESTION = ("According to the common sense knowledge and possible visual cues, what is the most likely location of "
                "the {} in the image?")                                         # visual_search.py:428
ANSWER_TEMPLATE = "Sure, [LOC]."                                                # VisualSearch/utils/utils.py:18-20


def background_color() -> Tuple[int, int, int]:
    return tuple(int(x * 255) for x in CLIP_MEAN)  # (122, 116, 104), visual_search.py:186


def expand2square(img: Image.Image, color=None) -> Image.Image:
    color = background_color() if color is None else color
    w, h = img.size
    if w == h:
        return img
    side = max(w, h)
    out = Image.new(img.mode, (side, side), color)
    out.paste(img, (0, 0))
    return out


def _normalise(arr_u8: np.ndarray) -> np.ndarray:
    # HF rescale multiplies in float64 and casts to float32, then normalises in float32 (image_transforms.rescale/normalize)
    x = (arr_u8.astype(np.float64) * (1 / 255)).astype(np.float32)
    x = (x - np.asarray(CLIP_MEAN, np.float32)) / np.asarray(CLIP_STD, np.float32)
    return np.ascontiguousarray(x.transpose(2, 0, 1))


def clip_preprocess(img: Image.Image, size: int = 224) -> np.ndarray:
    """CLIPImageProcessor.preprocess(expand2square(img)) -> float32 [3, size, size]."""
    img = expand2square(img.convert("RGB"))
    w, h = img.size
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long / short)
    nw, nh = (new_short, new_long) if w <= h else (new_long, new_short)
    img = img.resize((nw, nh), resample=Image.BICUBIC)
    left, top = (nw - size) // 2, (nh - size) // 2
    img = img.crop((left, top, left + size, top + size))
    return _normalise(np.asarray(img))


def owl_preprocess(img: Image.Image, size: int = 768) -> np.ndarray:
    """OwlViTProcessor(images=np.array(img)) -> float32 [3, size, size]."""
    img = img.convert("RGB").resize((size, size), resample=Image.BICUBIC)
    return _normalise(np.asarray(img))


# conv_templates["llava_llama_2"] (conversation.py:300-311), SeparatorStyle.LLAMA_2 (:72-93)
LLAVA_LLAMA_2_SYSTEM = ("You are a helpful language and vision assistant. You are able to understand the visual content that the "
                        "user provides, and assist the user with a variety of tasks using natural language.")


def build_prompt(question: str, use_mm_start_end: bool = True, answer: str | None = None, conv_type: str = "llava_v1") -> str:
    """The prompt VSM.inference builds (visual_search.py:176-184) for `--conv_type` llava_v1 (default) or llava_llama_2;
    `answer` teacher-forces the reply (without the closing </s>).  Pinned to the reference's templates by
    tests/golden/prompts.json (oracle/gen_prompt_golden.py)."""
    prompt = DEFAULT_IMAGE_TOKEN + "\n" + question
    if use_mm_start_end:
        prompt = prompt.replace(DEFAULT_IMAGE_TOKEN, DEFAULT_IM_START_TOKEN + DEFAULT_IMAGE_TOKEN + DEFAULT_IM_END_TOKEN)
    if conv_type == "llava_llama_2":
        ret = "[INST] <<SYS>>\n" + LLAVA_LLAMA_2_SYSTEM + "\n<</SYS>>\n\n" + prompt + " [/INST]"
        return ret + (" " + answer if answer else "")
    if conv_type != "llava_v1":
        raise ValueError(f"unknown conv_type {conv_type!r} (visual_search.py:47 offers llava_v1 and llava_llama_2)")
    ret = LLAVA_V1_SYSTEM + LLAVA_V1_SEP
    ret += LLAVA_V1_ROLES[0] + ": " + prompt + LLAVA_V1_SEP
    if answer:
        ret += LLAVA_V1_ROLES[1] + ": " + answer
    else:
        ret += LLAVA_V1_ROLES[1] + ":"
    return ret


def tokenizer_image_token(prompt: str, tokenizer, image_token_index: int = IMAGE_TOKEN_INDEX) -> List[int]:
    """Splits on <image>, tokenises each chunk, joins with the -200 placeholder and keeps a single BOS
    (VisualSearch/model/llava/mm_utils.py:19-44)."""
    chunks = [tokenizer(c).input_ids for c in prompt.split(DEFAULT_IMAGE_TOKEN)]
    ids: List[int] = []
    offset = 0
    bos = getattr(tokenizer, "bos_token_id", None)
    if chunks and chunks[0] and chunks[0][0] == bos:
        offset = 1
        ids.append(chunks[0][0])
    for i, c in enumerate(chunks):
        if i > 0:
            ids.extend([image_token_index] * 1)
        ids.extend(c[offset:])
    return ids


class SyntheticTokenizer:
    """Deterministic stand-in used when no tokenizer files are staged (this environment has no network): BOS=1, EOS=2,
    specials <im_start>, <im_end>, [LOC] at the top of the vocabulary (where `add_tokens` puts them for
    craigwu/seal_vsm_7b), every other whitespace/punctuation-delimited piece hashed into [3, vocab-4)."""

    def __init__(self, vocab_size: int = 32004):
        self.vocab_size = vocab_size
        self.bos_token_id, self.eos_token_id, self.unk_token_id = 1, 2, 0
        self.special = {DEFAULT_IM_START_TOKEN: vocab_size - 3, DEFAULT_IM_END_TOKEN: vocab_size - 2, "[LOC]": vocab_size - 1}

    def _pieces(self, text: str) -> List[str]:
        import re
        pat = "(" + "|".join(re.escape(s) for s in self.special) + r"|\w+|[^\w\s])"
        return [p for p in re.findall(pat, text) if p.strip()]

    def _id(self, piece: str) -> int:
        if piece in self.special:
            return self.special[piece]
        h = 2166136261
        for ch in piece.encode():
            h = ((h ^ ch) * 16777619) & 0xFFFFFFFF
        return 3 + h % (self.vocab_size - 3 - 4)

    def __call__(self, text: str, add_special_tokens: bool = True):
        ids = [self._id(p) for p in self._pieces(text)]
        if add_special_tokens:
            ids = [self.bos_token_id] + ids

        class _Enc:
            pass

        e = _Enc()
        e.input_ids = ids
        return e

    def batch_decode(self, seqs, skip_special_tokens: bool = True):
        return ["<synthetic:%d tokens>" % len(s) for s in seqs]
