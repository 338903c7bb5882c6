"""Model geometry of the SEAL visual-search model (VSM), mirrored 1:1 by `vstar_config` in include/vstar_hip.h.

Reference sources of the numbers (paths relative to the reference repo):
  * LLaVA/LLaMA config .......... VisualSearch/model/llava/model/language_model/llava_llama.py:31-52 (vicuna-7b defaults)
  * CLIP-ViT-L/14 tower ......... VisualSearch/model/llava/model/multimodal_encoder/clip_encoder.py:17-29, select_layer=-2
  * OWL-ViT-B/16 @ 768 .......... VisualSearch/model/owlvit/owlvit.py:21-31
  * SAM-style head (fixed) ...... VisualSearch/model/VSM.py:91-113
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass, asdict

ABI_VERSION = 1
IMAGE_TOKEN_INDEX = -200  # VisualSearch/utils/utils.py:7-12
N_BOXES = 2304
MASK_RES = 192
MAX_VERIFY = 8


class CVstarConfig(ctypes.Structure):
    _fields_ = [
        ("abi_version", ctypes.c_int32),
        ("clip_image_size", ctypes.c_int32),
        ("clip_patch", ctypes.c_int32),
        ("clip_hidden", ctypes.c_int32),
        ("clip_heads", ctypes.c_int32),
        ("clip_mlp", ctypes.c_int32),
        ("clip_layers", ctypes.c_int32),
        ("clip_select_layer", ctypes.c_int32),
        ("llm_hidden", ctypes.c_int32),
        ("llm_heads", ctypes.c_int32),
        ("llm_mlp", ctypes.c_int32),
        ("llm_layers", ctypes.c_int32),
        ("llm_vocab", ctypes.c_int32),
        ("llm_rms_eps", ctypes.c_float),
        ("llm_rope_theta", ctypes.c_float),
        ("owl_image_size", ctypes.c_int32),
        ("owl_patch", ctypes.c_int32),
        ("owl_hidden", ctypes.c_int32),
        ("owl_heads", ctypes.c_int32),
        ("owl_mlp", ctypes.c_int32),
        ("owl_layers", ctypes.c_int32),
        ("owl_query_dim", ctypes.c_int32),
        ("max_batch", ctypes.c_int32),
        ("max_text_len", ctypes.c_int32),
        ("llm_w8a8", ctypes.c_int32),
        ("reserved", ctypes.c_int32 * 7),
    ]


@dataclass
class VSMConfig:
    clip_image_size: int = 224
    clip_patch: int = 14
    clip_hidden: int = 1024
    clip_heads: int = 16
    clip_mlp: int = 4096
    clip_layers: int = 24
    clip_select_layer: int = -2
    llm_hidden: int = 4096
    llm_heads: int = 32
    llm_mlp: int = 11008
    llm_layers: int = 32
    llm_vocab: int = 32004
    llm_rms_eps: float = 1e-6
    llm_rope_theta: float = 10000.0
    owl_image_size: int = 768
    owl_patch: int = 16
    owl_hidden: int = 768
    owl_heads: int = 12
    owl_mlp: int = 3072
    owl_layers: int = 12
    owl_query_dim: int = 512
    max_batch: int = 32
    max_text_len: int = 192
    llm_w8a8: int = 0               # 1: LLaMA linears on the fp8 MFMA (BASELINE config 5); default bf16

    # ---- derived ----
    @property
    def n_img_tokens(self) -> int:
        return (self.clip_image_size // self.clip_patch) ** 2

    @property
    def clip_blocks(self) -> int:
        return self.clip_layers + 1 + self.clip_select_layer

    def to_c(self) -> CVstarConfig:
        c = CVstarConfig()
        c.abi_version = ABI_VERSION
        for k, v in asdict(self).items():
            setattr(c, k, v)
        return c

    @classmethod
    def seal_7b(cls, image_size: int = 224, **kw) -> "VSMConfig":
        """craigwu/seal_vsm_7b geometry; image_size 224 = reference, 336 = benchmark geometry (BASELINE.json)."""
        return cls(clip_image_size=image_size, **kw)

    @classmethod
    def tiny(cls, **kw) -> "VSMConfig":
        """Small-width model with the real topology (head dims 64/128, 224 px CLIP, 768 px OWL-ViT, full SAM head);
        used by the golden fixtures generated from the reference (oracle/gen_golden.py)."""
        d = dict(clip_hidden=128, clip_heads=2, clip_mlp=256, clip_layers=3, llm_hidden=256, llm_heads=2, llm_mlp=512,
                 llm_layers=2, llm_vocab=320, owl_hidden=128, owl_heads=2, owl_mlp=256, owl_layers=2, max_batch=4,
                 max_text_len=32)
        d.update(kw)
        return cls(**d)

    # ---- algorithmic work: FLOPs = 2*MACs, causal attention halved, 1 lm_head row ----
    def flops_per_crop(self, text_tokens: int = 64, full: bool = True) -> dict:
        P = self.n_img_tokens
        N = P + 1
        S = P + text_tokens
        Hc, Mc = self.clip_hidden, self.clip_mlp
        clip = 2 * P * (3 * self.clip_patch ** 2) * Hc
        clip += self.clip_blocks * (2 * N * Hc * (4 * Hc + 2 * Mc) + 4 * N * N * Hc)
        proj = 2 * P * Hc * self.llm_hidden
        H, M = self.llm_hidden, self.llm_mlp
        llm_lin = self.llm_layers * 2 * S * H * (4 * H + 3 * M)
        llm_att = self.llm_layers * 2 * S * S * H  # causal: half of 4*S*S*H
        head = 2 * H * self.llm_vocab + 2 * (2 * H * H + H * (self.owl_query_dim + 256))
        out = {"clip": clip, "projector": proj, "llm_linear": llm_lin, "llm_attention": llm_att, "llm_head": head}
        core = clip + proj + llm_lin + llm_att + head
        out["core"] = core
        if full:
            Ho, Mo = self.owl_hidden, self.owl_mlp
            Po = (self.owl_image_size // self.owl_patch) ** 2
            No = Po + 1
            owl = 2 * Po * (3 * self.owl_patch ** 2) * Ho
            owl += self.owl_layers * (2 * No * Ho * (4 * Ho + 2 * Mo) + 4 * No * No * Ho)
            heads = 2 * Po * Ho * (self.owl_query_dim + 2) + 2 * Po * Ho * (2 * Ho + 4) + 2 * Po * Ho * 256
            sam = 2 * (96 * 96) * 2304 * 64 + 2 * (192 * 192) * 576 * 32 + 1.5e9
            out.update({"owl_tower": owl, "det_heads": heads, "sam_head": sam})
            out["full"] = core + owl + heads + sam
        return out


# ------------------------------------------------------------------------------------------------------------
# VQA-LLM: LlavaSearchLlamaForCausalLM, mirrored 1:1 by `vstar_vqa_config` in include/vstar_vqa.h
#   LlavaSearchConfig(LlamaConfig) ...... LLaVA/llava/model/language_model/llava_search_llama.py:30-50
#   projector builder ................... LLaVA/llava/model/multimodal_projector/builder.py:33-68 (perceiver: depth 6,
#                                         16 heads x 96, 32 latents, 1 media embedding)
# ------------------------------------------------------------------------------------------------------------
VQA_ABI_VERSION = 1
OBJECT_TOKEN_INDEX = -300  # LLaVA/llava/constants.py:10
PAD_ROW = -(2 ** 31)


class CVqaConfig(ctypes.Structure):
    _fields_ = [
        ("abi_version", ctypes.c_int32),
        ("clip_image_size", ctypes.c_int32), ("clip_patch", ctypes.c_int32), ("clip_hidden", ctypes.c_int32),
        ("clip_heads", ctypes.c_int32), ("clip_mlp", ctypes.c_int32), ("clip_layers", ctypes.c_int32),
        ("clip_select_layer", ctypes.c_int32),
        ("llm_hidden", ctypes.c_int32), ("llm_heads", ctypes.c_int32), ("llm_mlp", ctypes.c_int32),
        ("llm_layers", ctypes.c_int32), ("llm_vocab", ctypes.c_int32),
        ("llm_rms_eps", ctypes.c_float), ("llm_rope_theta", ctypes.c_float),
        ("projector_type", ctypes.c_int32),
        ("pcv_depth", ctypes.c_int32), ("pcv_heads", ctypes.c_int32), ("pcv_dim_head", ctypes.c_int32),
        ("pcv_latents", ctypes.c_int32), ("pcv_ff_mult", ctypes.c_int32),
        ("max_slots", ctypes.c_int32), ("max_ctx", ctypes.c_int32), ("max_rows", ctypes.c_int32),
        ("max_images", ctypes.c_int32),
        ("reserved", ctypes.c_int32 * 8),
    ]


@dataclass
class VQAConfig:
    clip_image_size: int = 224
    clip_patch: int = 14
    clip_hidden: int = 1024
    clip_heads: int = 16
    clip_mlp: int = 4096
    clip_layers: int = 24
    clip_select_layer: int = -2
    llm_hidden: int = 4096
    llm_heads: int = 32
    llm_mlp: int = 11008
    llm_layers: int = 32
    llm_vocab: int = 32001          # vicuna tokenizer + <im_patch> (builder.py:131-135)
    llm_rms_eps: float = 1e-5
    llm_rope_theta: float = 10000.0
    projector_type: int = 0         # 0 linear, 1 mlp2x_gelu
    pcv_depth: int = 6
    pcv_heads: int = 16
    pcv_dim_head: int = 96
    pcv_latents: int = 32
    pcv_ff_mult: int = 4
    max_slots: int = 16
    max_ctx: int = 2048
    max_rows: int = 8192
    max_images: int = 16

    @property
    def n_img_tokens(self) -> int:
        return (self.clip_image_size // self.clip_patch) ** 2

    @property
    def feat_rows(self) -> int:
        """Rows of one feature-table slot: P long rows then pcv_latents short rows."""
        return self.n_img_tokens + self.pcv_latents

    def to_c(self) -> CVqaConfig:
        c = CVqaConfig()
        c.abi_version = VQA_ABI_VERSION
        for k, v in asdict(self).items():
            setattr(c, k, v)
        return c

    @classmethod
    def seal_7b(cls, **kw) -> "VQAConfig":
        """craigwu/seal_vqa_7b geometry (vicuna-7b + CLIP-L/14@224 + perceiver object projector)."""
        return cls(**kw)

    @classmethod
    def tiny(cls, **kw) -> "VQAConfig":
        """Small widths, real topology (head dims 64 / 128 / 96-wide perceiver heads, 224 px CLIP): golden fixtures."""
        d = dict(clip_hidden=128, clip_heads=2, clip_mlp=256, clip_layers=3, llm_hidden=256, llm_heads=2, llm_mlp=512,
                 llm_layers=2, llm_vocab=320, pcv_depth=2, pcv_heads=2, pcv_dim_head=96, pcv_latents=32, max_slots=8,
                 max_ctx=1024, max_rows=2048, max_images=8)
        d.update(kw)
        return cls(**d)
