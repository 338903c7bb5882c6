"""Scheduler oracle — TEST INFRASTRUCTURE ONLY.

(1) `FakeVSM`: a deterministic stand-in for the VSM (a pure function of the crop's size and pixels' checksum) so the
    search scheduler can be exercised on CPU without the 7B model.
(2) `load_reference_search()`: imports the REFERENCE's own visual_search.py (build container only; spaCy / cv2 /
    matplotlib / tqdm and the model imports are stubbed out, no reference source is copied) so that
    oracle/gen_search_golden.py can record the reference's search paths for the committed fixtures
    tests/golden/search_paths.json.
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

REF = os.environ.get("VSTAR_REFERENCE", "/root/reference")


class FakeVSM:
    """inference(image, question, mode) -> same conventions as visual_search.py:208-225."""

    def __init__(self, seed: int = 0, n_boxes: int = 64, gain: float = 9.0, conf_shift: float = -2.5, vqa_text=None,
                 symmetric: bool = False):
        self.seed, self.n_boxes, self.gain, self.conf_shift = seed, n_boxes, gain, conf_shift
        self.vqa_text = vqa_text
        # symmetric: the low-res map is mirrored left-right, so sibling sub-patches carry (mathematically) EQUAL heat mass and
        # the reference's order among them is decided by float32 rounding — the near-tie regime of the scheduler
        self.symmetric = symmetric
        self.calls = 0
        self.questions = []

    def _low(self, g, scale):
        low = torch.randn(1, 1, 12, 12, generator=g) * scale
        if self.symmetric:
            low[..., 6:] = torch.flip(low[..., :6], dims=[-1])
        return low

    def _rng(self, image):
        w, h = image.size
        px = np.asarray(image.resize((8, 8))).astype(np.int64).sum()
        return torch.Generator().manual_seed(int((self.seed * 1_000_003 + w * 7919 + h * 104729 + px) % (2 ** 31)))

    def inference(self, image, question, mode="segmentation"):
        self.calls += 1
        self.questions.append((mode, question))
        g = self._rng(image)
        w, h = image.size
        if mode == "vqa":
            if self.vqa_text is None:
                raise NotImplementedError
            return self.vqa_text
        low = self._low(g, self.gain if mode == "detection" else 9.0)
        heat = torch.clamp(F.interpolate(low, (h, w), mode="bilinear", align_corners=False)[0, 0], min=0)
        if mode == "segmentation":
            return heat
        boxes = torch.rand(self.n_boxes, 4, generator=g)
        scores = torch.sigmoid(torch.randn(self.n_boxes, 1, generator=g) * 1.5 + self.conf_shift)
        return boxes, scores, heat


def load_reference_search():
    """Returns the reference's `visual_search` module object (functions visual_search, visual_search_queue, ...)."""
    assert os.path.isdir(REF), "reference tree not present"
    import importlib.machinery

    def stub(name, **attrs):
        if name in sys.modules:
            return sys.modules[name]
        m = types.ModuleType(name)
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    stub("spacy", load=lambda *_a, **_k: (lambda text: []))
    stub("cv2", COLORMAP_JET=2)
    stub("tqdm", tqdm=lambda x, *a, **k: x)
    mpl = stub("matplotlib")
    mpl.pyplot = stub("matplotlib.pyplot")
    vs = stub("VisualSearch"); vs.__path__ = []
    stub("VisualSearch.model").__path__ = []
    stub("VisualSearch.model.VSM", VSMForCausalLM=object)
    stub("VisualSearch.model.llava", conversation=types.SimpleNamespace(conv_templates={})).__path__ = []
    stub("VisualSearch.model.llava.conversation", conv_templates={})
    stub("VisualSearch.model.llava.mm_utils", tokenizer_image_token=None)
    stub("VisualSearch.utils").__path__ = []
    stub("VisualSearch.utils.utils", expand2square=None, DEFAULT_IM_END_TOKEN="<im_end>", DEFAULT_IM_START_TOKEN="<im_start>",
         DEFAULT_IMAGE_TOKEN="<image>", IMAGE_TOKEN_INDEX=-200)
    import importlib.util
    spec = importlib.util.spec_from_file_location("_ref_visual_search", os.path.join(REF, "visual_search.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod
