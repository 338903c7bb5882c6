"""Imports the REFERENCE VQA-LLM (`LLaVA/llava/...` under /root/reference, read-only) on CPU — build-container only.

TEST INFRASTRUCTURE.  Same recipe as oracle/ref_shim.py: package stubs that skip the reference's
`__init__.py` side effects, `register(..., exist_ok=True)`, a local CLIP config instead of a hub download, plus a
two-function stand-in for the absent `einops_exts` package (rearrange_many / repeat_many = map over einops).  No
reference source is copied: the modules are imported from where they lie.  Used by oracle/gen_vqa_golden.py to pin
oracle/vqa_oracle.py and to generate tests/golden/vqa_*.npz; nothing that runs on the GPU box imports this file.
"""
from __future__ import annotations

import os
import sys
import types

import torch

REF = os.environ.get("VSTAR_REFERENCE", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REF, "LLaVA", "llava", "model"))


def load_reference(cfg):
    """Builds the reference's LlavaSearchLlamaForCausalLM (random init, eval) for a vstar_amd.config.VQAConfig."""
    import transformers  # noqa: F401
    from transformers import CLIPVisionConfig, CLIPVisionModel

    def stub(name, rel):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [REF + rel]
            sys.modules[name] = m

    for n, p in [("LLaVA", "/LLaVA"), ("LLaVA.llava", "/LLaVA/llava"), ("LLaVA.llava.model", "/LLaVA/llava/model"),
                 ("LLaVA.llava.model.language_model", "/LLaVA/llava/model/language_model"),
                 ("LLaVA.llava.model.multimodal_encoder", "/LLaVA/llava/model/multimodal_encoder"),
                 ("LLaVA.llava.model.multimodal_projector", "/LLaVA/llava/model/multimodal_projector")]:
        stub(n, p)
    if "einops_exts" not in sys.modules:
        import einops
        ex = types.ModuleType("einops_exts")
        ex.rearrange_many = lambda ts, pattern, **kw: tuple(einops.rearrange(t, pattern, **kw) for t in ts)
        ex.repeat_many = lambda ts, pattern, **kw: tuple(einops.repeat(t, pattern, **kw) for t in ts)
        sys.modules["einops_exts"] = ex
    import transformers.models.auto.auto_factory as af
    import transformers.models.auto.configuration_auto as ca
    if not getattr(ca.AutoConfig, "_vstar_patched", False):
        _r = ca.AutoConfig.register
        ca.AutoConfig.register = staticmethod(lambda t, c, exist_ok=False: _r(t, c, exist_ok=True))
        _m = af._BaseAutoModelClass.register.__func__
        af._BaseAutoModelClass.register = classmethod(lambda cls, c, m, exist_ok=False: _m(cls, c, m, exist_ok=True))
        ca.AutoConfig._vstar_patched = True
    ccfg = CLIPVisionConfig(hidden_size=cfg.clip_hidden, intermediate_size=cfg.clip_mlp, num_hidden_layers=cfg.clip_layers,
                            num_attention_heads=cfg.clip_heads, patch_size=cfg.clip_patch, image_size=cfg.clip_image_size,
                            projection_dim=64)
    ccfg._attn_implementation = "eager"
    CLIPVisionConfig.from_pretrained = classmethod(lambda cls, *a, **k: ccfg)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import importlib
    M = importlib.import_module("LLaVA.llava.model.language_model.llava_search_llama")
    if cfg.pcv_depth != 6 or cfg.pcv_heads != 16 or cfg.pcv_latents != 32 or cfg.pcv_dim_head != 96:
        # builder.py:54-66 hard-codes PerceiverResampler(depth=6, heads=16, dim_head=96, num_latents=32); the tiny fixtures
        # use fewer layers/heads, so wrap the class (in memory) to override exactly those constructor arguments
        B = importlib.import_module("LLaVA.llava.model.multimodal_projector.builder")
        Orig = importlib.import_module("LLaVA.llava.model.multimodal_projector.perceiver").PerceiverResampler

        def make(**kw):
            kw.update(depth=cfg.pcv_depth, heads=cfg.pcv_heads, dim_head=cfg.pcv_dim_head, num_latents=cfg.pcv_latents)
            return Orig(**kw)
        B.PerceiverResampler = make
    lc = M.LlavaSearchConfig(vocab_size=cfg.llm_vocab, hidden_size=cfg.llm_hidden, intermediate_size=cfg.llm_mlp,
                             num_hidden_layers=cfg.llm_layers, num_attention_heads=cfg.llm_heads,
                             num_key_value_heads=cfg.llm_heads, rms_norm_eps=cfg.llm_rms_eps, max_position_embeddings=2048,
                             rope_theta=cfg.llm_rope_theta)
    lc._attn_implementation = "eager"
    lc.mm_vision_tower = "openai/clip-vit-large-patch14"
    lc.mm_hidden_size = cfg.clip_hidden
    lc.mm_vision_select_layer = cfg.clip_select_layer
    lc.mm_vision_select_feature = "patch"
    lc.mm_projector_type = "linear" if cfg.projector_type == 0 else "mlp2x_gelu"
    lc.object_mm_projector_type = "perceiver"
    lc.use_cache = True
    model = M.LlavaSearchLlamaForCausalLM(lc).eval()
    vt = model.get_model().get_vision_tower()
    vt.vision_tower = CLIPVisionModel(ccfg).eval()
    vt.is_loaded = True
    return M, model


def load_state(model, sd):
    """Copies an engine-keyed state dict (vstar_amd.weights.vqa_state_dict_spec) into the reference model."""
    own = model.state_dict()
    clip_own = model.get_model().get_vision_tower().vision_tower.state_dict()
    missing = []
    with torch.no_grad():
        for k, v in sd.items():
            if k.startswith("clip."):
                kk = k[len("clip."):]
                if kk not in clip_own and kk.startswith("vision_model."):
                    kk = kk[len("vision_model."):]
                if kk in clip_own:
                    clip_own[kk].copy_(v.to(clip_own[kk].dtype))
                else:
                    missing.append(k)
            elif k in own:
                own[k].copy_(v.to(own[k].dtype).reshape(own[k].shape))
            else:
                missing.append(k)
    return missing
