"""Imports the REFERENCE implementation (read-only tree at /root/reference) on CPU — build-container only.

TEST INFRASTRUCTURE.  /root/reference does not exist on the GPU box; nothing under tests -m gpu, smoke() or bench.py
imports this file.  It exists to (a) pin oracle/vsm_oracle.py against the reference's own `model_forward(inference=True)`
and (b) generate the golden vectors committed under tests/golden/ (see oracle/gen_golden.py).

The reference cannot be imported as-is in this image (spacy/cv2/torchvision/peft absent, transformers 5.x instead of
4.31).  The recipe below registers package stubs that skip the reference's `__init__.py` side
effects, a 3-function torchvision stub, `register(..., exist_ok=True)`, local HF configs instead of hub downloads, and
`Tensor.cuda -> identity`.  No reference source is copied: modules are imported from where they lie.
"""
from __future__ import annotations

import os
import sys
import types

import torch

REF = os.environ.get("VSTAR_REFERENCE", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REF, "VisualSearch", "model"))


def load_reference(cfg, loc_token_idx: int):
    """Builds the reference's VSMForCausalLM (random init) for a vstar_amd.config.VSMConfig and returns (module V, model)."""
    import transformers  # noqa: F401  (must be imported before the torchvision stub)
    from transformers import CLIPVisionConfig, CLIPVisionModel, OwlViTConfig

    def stub(name, rel):
        m = types.ModuleType(name)
        m.__path__ = [REF + rel]
        sys.modules[name] = m

    for n, p in [("VisualSearch", "/VisualSearch"), ("VisualSearch.model", "/VisualSearch/model"),
                 ("VisualSearch.model.llava", "/VisualSearch/model/llava"),
                 ("VisualSearch.model.llava.model", "/VisualSearch/model/llava/model"),
                 ("VisualSearch.model.llava.model.language_model", "/VisualSearch/model/llava/model/language_model"),
                 ("VisualSearch.model.segment_anything", "/VisualSearch/model/segment_anything")]:
        if n not in sys.modules:
            stub(n, p)
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tv.__version__ = "0.16.1"
        ops = types.ModuleType("torchvision.ops")
        bx = types.ModuleType("torchvision.ops.boxes")
        ms = types.ModuleType("torchvision.ops.misc")
        bx.box_area = lambda b: (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
        ops.boxes = bx
        ops.misc = ms
        tv.ops = ops
        sys.modules.update({"torchvision": tv, "torchvision.ops": ops, "torchvision.ops.boxes": bx,
                            "torchvision.ops.misc": ms})
    import transformers.models.auto.auto_factory as af
    import transformers.models.auto.configuration_auto as ca
    if not getattr(ca.AutoConfig, "_vstar_patched", False):
        _r = ca.AutoConfig.register
        ca.AutoConfig.register = staticmethod(lambda t, c, exist_ok=False: _r(t, c, exist_ok=True))
        _m = af._BaseAutoModelClass.register.__func__
        af._BaseAutoModelClass.register = classmethod(lambda cls, c, m, exist_ok=False: _m(cls, c, m, exist_ok=True))
        ca.AutoConfig._vstar_patched = True

    ocfg = OwlViTConfig(vision_config=dict(hidden_size=cfg.owl_hidden, intermediate_size=cfg.owl_mlp,
                                           num_hidden_layers=cfg.owl_layers, num_attention_heads=cfg.owl_heads,
                                           patch_size=cfg.owl_patch, image_size=cfg.owl_image_size),
                        text_config=dict(hidden_size=cfg.owl_query_dim))
    ccfg = CLIPVisionConfig(hidden_size=cfg.clip_hidden, intermediate_size=cfg.clip_mlp,
                            num_hidden_layers=cfg.clip_layers, num_attention_heads=cfg.clip_heads,
                            patch_size=cfg.clip_patch, image_size=cfg.clip_image_size, projection_dim=64)
    ocfg._attn_implementation = "eager"
    ccfg._attn_implementation = "eager"
    OwlViTConfig.from_pretrained = classmethod(lambda cls, *a, **k: ocfg)
    CLIPVisionConfig.from_pretrained = classmethod(lambda cls, *a, **k: ccfg)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.cuda.empty_cache = lambda: None

    if REF not in sys.path:
        sys.path.insert(0, REF)
    import importlib
    P = cfg.n_img_tokens
    if P == 256:
        V = importlib.import_module("VisualSearch.model.VSM")
    else:
        # the reference hard-codes 255 = 256-1 image tokens (VSM.py:232,469); for other CLIP geometries exec a
        # source-patched module object (in memory only) with the literal generalised to P-1
        src = open(REF + "/VisualSearch/model/VSM.py").read()
        n = src.count(", 255))")
        assert n == 2, n
        src = src.replace(", 255))", f", {P - 1}))")
        V = types.ModuleType("VisualSearch.model.VSM_patched")
        V.__package__ = "VisualSearch.model"
        V.__file__ = REF + "/VisualSearch/model/VSM.py"
        exec(compile(src, V.__file__, "exec"), V.__dict__)
    from VisualSearch.model.llava.model.language_model.llava_llama import LlavaConfig

    lc = LlavaConfig(vocab_size=cfg.llm_vocab, hidden_size=cfg.llm_hidden, intermediate_size=cfg.llm_mlp,
                     num_hidden_layers=cfg.llm_layers, num_attention_heads=cfg.llm_heads,
                     num_key_value_heads=cfg.llm_heads, rms_norm_eps=cfg.llm_rms_eps, max_position_embeddings=2048)
    lc._attn_implementation = "eager"
    lc.mm_vision_tower = lc.vision_tower = "openai/clip-vit-large-patch14"
    lc.mm_hidden_size = cfg.clip_hidden
    lc.mm_vision_select_layer = cfg.clip_select_layer
    lc.mm_use_im_start_end = True
    lc.train_mask_decoder = True
    lc.out_dim = cfg.owl_query_dim
    model = V.VSMForCausalLM(lc, loc_token_idx=loc_token_idx, is_eval=True).eval()
    vt = model.get_model().get_vision_tower()
    vt.vision_tower = CLIPVisionModel(ccfg)
    vt.is_loaded = True
    return V, model


def load_state(model, sd):
    """Copies an engine-keyed state dict (vstar_amd.weights) into the reference model; returns the unmatched keys."""
    own = model.state_dict()
    vt = model.get_model().get_vision_tower().vision_tower
    clip_own = vt.state_dict()
    missing, used = [], set()
    with torch.no_grad():
        for k, v in sd.items():
            if k.startswith("clip."):
                kk = k[len("clip."):]
                if kk not in clip_own and kk.startswith("vision_model."):
                    kk = kk[len("vision_model."):]      # transformers>=5 flattens CLIPVisionModel's key prefix
                if kk in clip_own:
                    clip_own[kk].copy_(v.to(clip_own[kk].dtype))
                    used.add(k)
                else:
                    missing.append(k)
            elif k in own:
                own[k].copy_(v.to(own[k].dtype).reshape(own[k].shape))
                used.add(k)
            else:
                missing.append(k)
    return missing


def reference_forward(model, images_clip, images, input_ids, mask_hw=(192, 192)):
    """One crop through the reference's own model_forward(inference=True) (VSM.py:201-364)."""
    with torch.no_grad():
        return model.model_forward(
            images=images, images_clip=images_clip, input_ids=input_ids, labels=None,
            attention_masks=torch.ones_like(input_ids, dtype=torch.bool), offset=torch.tensor([0, 1]), masks_list=[],
            label_list=[torch.zeros(mask_hw)], bboxes_labels_list=[], bboxes_valid_list=[], masks_valid_list=[],
            resize_list=[], inference=True)
