"""Checkpoint key map of the VSM (what `VSMForCausalLM.from_pretrained` + the separately fetched CLIP tower hold,
visual_search.py:157-161; key list in SURVEY.md §5) and a deterministic synthetic initialiser of that exact key set.

There is no network in the build/bench environment, so benchmarks and parity fixtures use seeded random weights of
the real architecture; `load_checkpoint_dir` is the path for a real `craigwu/seal_vsm_7b` + `openai/clip-vit-large-patch14`
checkpoint when one is staged locally.
"""
from __future__ import annotations

import math
import os
import zlib
from collections import OrderedDict
from typing import Dict, Iterable, Tuple

import torch

from .config import VSMConfig, VQAConfig

CLIP_PREFIX = "clip."  # the CLIP tower is NOT part of the VSM checkpoint (merge_lora_weights_and_save_hf_model.py:145-150)


def _vit_keys(prefix: str, preln: str, hidden: int, mlp: int, layers: int, patch: int, n_tokens: int, post: bool):
    e = prefix + "embeddings."
    yield e + "class_embedding", (hidden,)
    yield e + "patch_embedding.weight", (hidden, 3, patch, patch)
    yield e + "position_embedding.weight", (n_tokens, hidden)
    yield prefix + preln + ".weight", (hidden,)
    yield prefix + preln + ".bias", (hidden,)
    for i in range(layers):
        lp = f"{prefix}encoder.layers.{i}."
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            yield lp + f"self_attn.{nm}.weight", (hidden, hidden)
            yield lp + f"self_attn.{nm}.bias", (hidden,)
        yield lp + "layer_norm1.weight", (hidden,)
        yield lp + "layer_norm1.bias", (hidden,)
        yield lp + "mlp.fc1.weight", (mlp, hidden)
        yield lp + "mlp.fc1.bias", (mlp,)
        yield lp + "mlp.fc2.weight", (hidden, mlp)
        yield lp + "mlp.fc2.bias", (hidden,)
        yield lp + "layer_norm2.weight", (hidden,)
        yield lp + "layer_norm2.bias", (hidden,)
    if post:
        yield prefix + "post_layernorm.weight", (hidden,)
        yield prefix + "post_layernorm.bias", (hidden,)


def _sam_attn_keys(prefix: str, internal: int):
    for nm, shp in (("q_proj", (internal, 256)), ("k_proj", (internal, 256)), ("v_proj", (internal, 256)),
                    ("out_proj", (256, internal))):
        yield prefix + nm + ".weight", shp
        yield prefix + nm + ".bias", (shp[0],)


def state_dict_spec(cfg: VSMConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    """Every tensor the engine consumes, with its shape, in a fixed order (the order seeds the synthetic init)."""
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    P = cfg.n_img_tokens
    # CLIP tower (all 24 layers + post_layernorm exist in the checkpoint; the engine uses clip_blocks of them)
    for k, v in _vit_keys(CLIP_PREFIX + "vision_model.", "pre_layrnorm", cfg.clip_hidden, cfg.clip_mlp, cfg.clip_layers,
                          cfg.clip_patch, P + 1, post=True):
        s[k] = v
    H, M = cfg.llm_hidden, cfg.llm_mlp
    s["model.embed_tokens.weight"] = (cfg.llm_vocab, H)
    for i in range(cfg.llm_layers):
        lp = f"model.layers.{i}."
        for nm in ("q_proj", "k_proj", "v_proj", "o_proj"):
            s[lp + f"self_attn.{nm}.weight"] = (H, H)
        s[lp + "mlp.gate_proj.weight"] = (M, H)
        s[lp + "mlp.up_proj.weight"] = (M, H)
        s[lp + "mlp.down_proj.weight"] = (H, M)
        s[lp + "input_layernorm.weight"] = (H,)
        s[lp + "post_attention_layernorm.weight"] = (H,)
    s["model.norm.weight"] = (H,)
    s["lm_head.weight"] = (cfg.llm_vocab, H)
    s["model.mm_projector.weight"] = (H, cfg.clip_hidden)
    s["model.mm_projector.bias"] = (H,)
    # OWL-ViT
    Po = (cfg.owl_image_size // cfg.owl_patch) ** 2
    Ho = cfg.owl_hidden
    for k, v in _vit_keys("model.owlvit.vision_model.", "pre_layernorm", Ho, cfg.owl_mlp, cfg.owl_layers, cfg.owl_patch,
                          Po + 1, post=True):
        s[k] = v
    s["model.owlvit.layer_norm.weight"] = (Ho,)
    s["model.owlvit.layer_norm.bias"] = (Ho,)
    Q = cfg.owl_query_dim
    for nm, shp in (("dense0", (Q, Ho)), ("logit_shift", (1, Ho)), ("logit_scale", (1, Ho))):
        s[f"model.owlvit.class_head.{nm}.weight"] = shp
        s[f"model.owlvit.class_head.{nm}.bias"] = (shp[0],)
    for nm, shp in (("dense0", (Ho, Ho)), ("dense1", (Ho, Ho)), ("dense2", (4, Ho))):
        s[f"model.owlvit.box_head.{nm}.weight"] = shp
        s[f"model.owlvit.box_head.{nm}.bias"] = (shp[0],)
    # heads
    s["model.visual_projection.weight"] = (256, Ho)
    for br, od in (("det", Q), ("seg", 256)):
        s[f"model.text_hidden_fcs_{br}.0.0.weight"] = (H, H)
        s[f"model.text_hidden_fcs_{br}.0.0.bias"] = (H,)
        s[f"model.text_hidden_fcs_{br}.0.2.weight"] = (od, H)
        s[f"model.text_hidden_fcs_{br}.0.2.bias"] = (od,)
    # SAM prompt encoder / mask decoder
    s["model.prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"] = (2, 128)
    s["model.prompt_encoder.no_mask_embed.weight"] = (1, 256)
    md = "model.mask_decoder."
    s[md + "iou_token.weight"] = (1, 256)
    s[md + "mask_tokens.weight"] = (4, 256)
    for i in range(2):
        lp = f"{md}transformer.layers.{i}."
        for k, v in _sam_attn_keys(lp + "self_attn.", 256):
            s[k] = v
        for k, v in _sam_attn_keys(lp + "cross_attn_token_to_image.", 128):
            s[k] = v
        for k, v in _sam_attn_keys(lp + "cross_attn_image_to_token.", 128):
            s[k] = v
        for n in (1, 2, 3, 4):
            s[lp + f"norm{n}.weight"] = (256,)
            s[lp + f"norm{n}.bias"] = (256,)
        s[lp + "mlp.lin1.weight"] = (2048, 256)
        s[lp + "mlp.lin1.bias"] = (2048,)
        s[lp + "mlp.lin2.weight"] = (256, 2048)
        s[lp + "mlp.lin2.bias"] = (256,)
    for k, v in _sam_attn_keys(md + "transformer.final_attn_token_to_image.", 128):
        s[k] = v
    s[md + "transformer.norm_final_attn.weight"] = (256,)
    s[md + "transformer.norm_final_attn.bias"] = (256,)
    s[md + "output_upscaling.0.conv.weight"] = (64, 256, 3, 3)
    s[md + "output_upscaling.0.conv.bias"] = (64,)
    s[md + "output_upscaling.1.weight"] = (64,)
    s[md + "output_upscaling.1.bias"] = (64,)
    s[md + "output_upscaling.3.conv.weight"] = (32, 64, 3, 3)
    s[md + "output_upscaling.3.conv.bias"] = (32,)
    for j in range(3):
        s[md + f"output_hypernetworks_mlps.0.layers.{j}.weight"] = (256 if j < 2 else 32, 256)
        s[md + f"output_hypernetworks_mlps.0.layers.{j}.bias"] = (256 if j < 2 else 32,)
    return s


def _is_norm_weight(key: str) -> bool:
    k = key.rsplit(".", 1)[0]
    last = k.rsplit(".", 1)[-1]
    if key.endswith(".weight") and "mm_projector_object" in key:
        return "norm" in last or k.endswith("mm_projector_object.0") or k.endswith(".1.0")     # LayerNorm gains
    return key.endswith(".weight") and (
        "norm" in last or last == "pre_layrnorm" or k.endswith("output_upscaling.1"))


def vqa_state_dict_spec(cfg: VQAConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    """Every tensor the VQA-LLM engine consumes: HF keys of LlavaSearchLlamaForCausalLM (llava_search_llama.py:40-50,
    llava_search_arch.py:14-19, builder.py:33-68, perceiver.py:25-99) plus the separately loaded CLIP tower ("clip.")."""
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    P, C, H = cfg.n_img_tokens, cfg.clip_hidden, cfg.llm_hidden
    for k, v in _vit_keys(CLIP_PREFIX + "vision_model.", "pre_layrnorm", C, cfg.clip_mlp, cfg.clip_layers, cfg.clip_patch,
                          P + 1, post=True):
        s[k] = v
    if cfg.projector_type == 0:
        s["model.mm_projector.weight"] = (H, C)
        s["model.mm_projector.bias"] = (H,)
    else:
        s["model.mm_projector.0.weight"] = (H, C)
        s["model.mm_projector.0.bias"] = (H,)
        s["model.mm_projector.2.weight"] = (H, H)
        s["model.mm_projector.2.bias"] = (H,)
    po = "model.mm_projector_object."
    inner = cfg.pcv_heads * cfg.pcv_dim_head
    s[po + "0.weight"] = (C,)
    s[po + "0.bias"] = (C,)
    s[po + "1.latents"] = (cfg.pcv_latents, C)
    s[po + "1.media_pos_emb"] = (1, 1, C)
    for i in range(cfg.pcv_depth):
        lp = f"{po}1.layers.{i}."
        for nm in ("norm_media", "norm_latents"):
            s[lp + f"0.{nm}.weight"] = (C,)
            s[lp + f"0.{nm}.bias"] = (C,)
        s[lp + "0.to_q.weight"] = (inner, C)
        s[lp + "0.to_kv.weight"] = (2 * inner, C)
        s[lp + "0.to_out.weight"] = (C, inner)
        s[lp + "1.0.weight"] = (C,)
        s[lp + "1.0.bias"] = (C,)
        s[lp + "1.1.weight"] = (C * cfg.pcv_ff_mult, C)
        s[lp + "1.3.weight"] = (C, C * cfg.pcv_ff_mult)
    s[po + "1.norm.weight"] = (C,)
    s[po + "1.norm.bias"] = (C,)
    s[po + "2.weight"] = (H, C)
    s[po + "2.bias"] = (H,)
    s["model.embed_tokens.weight"] = (cfg.llm_vocab, H)
    for i in range(cfg.llm_layers):
        lp = f"model.layers.{i}."
        for nm in ("q_proj", "k_proj", "v_proj", "o_proj"):
            s[lp + f"self_attn.{nm}.weight"] = (H, H)
        s[lp + "mlp.gate_proj.weight"] = (cfg.llm_mlp, H)
        s[lp + "mlp.up_proj.weight"] = (cfg.llm_mlp, H)
        s[lp + "mlp.down_proj.weight"] = (H, cfg.llm_mlp)
        s[lp + "input_layernorm.weight"] = (H,)
        s[lp + "post_attention_layernorm.weight"] = (H,)
    s["model.norm.weight"] = (H,)
    s["lm_head.weight"] = (cfg.llm_vocab, H)
    return s


def random_state_dict(cfg, seed: int = 0, dtype: torch.dtype = torch.bfloat16,
                      keys: Iterable[str] | None = None, share_layers: bool = False) -> Dict[str, torch.Tensor]:
    """Seeded synthetic weights: norm gains 1+N(0,0.1), biases/embeddings N(0,0.02), matrices N(0, 1/fan_in).
    Each tensor draws from its own generator seeded by (seed, index) so any subset reproduces bit-identically.
    share_layers=True (throughput benchmarks only) reuses layer 0's host tensors for every other layer of a stack:
    the engine still packs and uploads one device copy per layer, so HBM footprint and traffic are unchanged."""
    import re
    spec = vqa_state_dict_spec(cfg) if isinstance(cfg, VQAConfig) else state_dict_spec(cfg)
    want = set(keys) if keys is not None else None
    out: Dict[str, torch.Tensor] = {}
    for idx, (k, shp) in enumerate(spec.items()):
        if want is not None and k not in want:
            continue
        if share_layers:
            m = re.search(r"\.layers\.(\d+)\.", k)
            if m and int(m.group(1)) > 0:
                k0 = k[:m.start()] + ".layers.0." + k[m.end():]
                if k0 in out and tuple(out[k0].shape) == tuple(shp):
                    out[k] = out[k0]
                    continue
        g = torch.Generator().manual_seed(seed * 1_000_003 + idx)
        if _is_norm_weight(k):
            t = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif k.endswith(".bias") or len(shp) == 1:
            t = 0.02 * torch.randn(shp, generator=g)
        elif "embed_tokens" in k or "position_embedding" in k or k.endswith("_token.weight") or k.endswith("_tokens.weight") \
                or "no_mask_embed" in k or k.endswith(".latents") or k.endswith("media_pos_emb"):
            t = 0.5 * torch.randn(shp, generator=g)
        elif "gaussian_matrix" in k:
            t = torch.randn(shp, generator=g)
        else:
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            t = torch.randn(shp, generator=g) / math.sqrt(fan_in)
        out[k] = t.to(dtype)
    return out


def template_chain(tokenizer, question: str = "Please locate the object in this image.", conv_type: str = "llava_v1",
                   use_mm_start_end: bool = True):
    """(token, next_token) pairs that make greedy decoding continue a locate prompt with "Sure, [LOC]." and then EOS:
    the last prompt token -> "Sure" -> "," -> "[LOC]" -> "." -> </s> for `tokenizer`'s ids (input of trained_like_state_dict)."""
    from .preprocess import ANSWER_TEMPLATE, build_prompt, tokenizer_image_token
    ids_p = tokenizer_image_token(build_prompt(question, use_mm_start_end, conv_type=conv_type), tokenizer)
    ids_f = tokenizer_image_token(build_prompt(question, use_mm_start_end, answer=ANSWER_TEMPLATE, conv_type=conv_type), tokenizer)
    if ids_f[:len(ids_p)] != ids_p:
        ids_f = ids_p + list(tokenizer(" " + ANSWER_TEMPLATE, add_special_tokens=False).input_ids)
    ids_f = list(ids_f) + [int(getattr(tokenizer, "eos_token_id", 2))]
    chain = [(int(ids_f[c - 1]), int(ids_f[c])) for c in range(len(ids_p), len(ids_f))]
    assert len({t for t, _ in chain}) == len(chain), "the answer template repeats a token: a bigram chain cannot encode it"
    return chain


def trained_like_state_dict(cfg: VSMConfig, seed: int = 0, dtype: torch.dtype = torch.bfloat16, share_layers: bool = True,
                            chain=None, features=("outliers", "attn", "norms")) -> Dict[str, torch.Tensor]:
    """Seeded weights with the STATISTICS of a trained checkpoint rather than i.i.d. N(0, s) (VERDICT r3 "missing" #2): the
    structures that decide where bf16 / fp8 rounding bites in a real LLaMA-7B / CLIP-L, which random_state_dict lacks.

      * outlier residual channels: a handful of output rows of every o_proj / down_proj (and out_proj / fc2 of the ViTs) are
        6 - 40 x stronger (25 x in the ViTs), so those channels of the residual stream run at tens to hundreds of times the rms of the rest;
      * a massive-activation token: BOS carries +-60 (120 x the embedding rms) in two of those channels (attention sink);
      * norm gains with a per-channel spread (log-normal around 0.4) and SMALL gains on the outlier channels, as trained
        LLaMA / CLIP norms have; LayerNorm biases of the ViTs with a heavier spread;
      * correlated q / k projections (k = 0.7 q + noise, q scaled x 3): attention is peaked instead of near-uniform;
      * `chain` [(token, next)]: the embeddings of the chain tokens are 8 x stronger (delimiter-like tokens) and lm_head rows
        are aligned with their predecessors' embeddings, so that greedy decoding emits the chain — with the pairs of
        `template_chain(tokenizer)` the model answers "Sure, [LOC]." like the trained VSM does and the DEFAULT
        strict_template=True path of VSM.inference runs at real widths.  The transformer layers stay
        non-trivial (unlike the zeroed-o/down bigram model of tests/test_template_fallback_gpu.py).

    `features` switches the three structures individually (error-attribution probes: tools/owl_error_probe.py); the default is all.
    Derived from random_state_dict(seed) tensor by tensor, so the key set / shapes / reproducibility rules are the same."""
    f_out, f_attn, f_norm = ("outliers" in features), ("attn" in features), ("norms" in features)
    sd = random_state_dict(cfg, seed=seed, dtype=torch.float32, share_layers=share_layers)
    g = torch.Generator().manual_seed(seed * 7919 + 17)
    done = set()

    def once(key):
        t = sd[key]
        if id(t) in done:
            return None
        done.add(id(t))
        return t

    def outliers(hidden, n, lo, hi):
        ch = torch.randperm(hidden, generator=g)[:n]
        amp = torch.logspace(math.log10(lo), math.log10(hi), n)
        return ch, amp

    def spread_gain(key, ch, centre, sigma, small):
        t = once(key)
        if t is not None:
            gk = torch.Generator().manual_seed(seed * 31 + (zlib.crc32(key.encode()) & 0xFFFFF))
            t.copy_(centre * torch.exp(sigma * torch.randn(t.shape, generator=gk)))
            t[ch] = small

    # ---- LLaMA ----
    H = cfg.llm_hidden
    ch, amp = outliers(H, max(2, H // 683), 6.0, 40.0)
    for i in range(cfg.llm_layers):
        lp = f"model.layers.{i}."
        for nm in ("self_attn.o_proj.weight", "mlp.down_proj.weight"):
            t = once(lp + nm)
            if t is not None and f_out:
                t[ch] *= amp[:, None]
        q, k = once(lp + "self_attn.q_proj.weight"), once(lp + "self_attn.k_proj.weight")
        if q is not None and k is not None and f_attn:
            k.copy_(0.7 * q + math.sqrt(1 - 0.49) * k)
            q *= 3.0
        if f_norm:
            spread_gain(lp + "input_layernorm.weight", ch, 0.4, 0.5, 0.05)
            spread_gain(lp + "post_attention_layernorm.weight", ch, 0.4, 0.5, 0.05)
    if f_norm:
        spread_gain("model.norm.weight", ch, 1.0, 0.3, 0.05)
    E = sd["model.embed_tokens.weight"]
    if f_out:
        E[1, ch[:2]] = torch.tensor([60.0, -60.0])[: len(ch[:2])]
    if chain:
        W = sd["lm_head.weight"]
        for t, _ in chain:
            E[t] *= 8.0
            E[t, ch] = 0.0
        for t, nxt in chain:
            W[nxt] += 6.0 * E[t] / E[t].norm()
    # ---- ViT towers ----
    for prefix, pre, hidden, layers in ((CLIP_PREFIX + "vision_model.", "pre_layrnorm", cfg.clip_hidden, cfg.clip_layers),
                                        ("model.owlvit.vision_model.", "pre_layernorm", cfg.owl_hidden, cfg.owl_layers)):
        vch, vamp = outliers(hidden, max(2, hidden // 256), 8.0, 25.0)
        if f_norm:
            spread_gain(prefix + pre + ".weight", vch, 1.0, 0.4, 0.3)
        for i in range(layers):
            lp = f"{prefix}encoder.layers.{i}."
            for nm in ("self_attn.out_proj", "mlp.fc2"):
                t = once(lp + nm + ".weight")
                if t is not None:
                    sgn = torch.sign(torch.randn(len(vch), generator=g))      # (drawn whatever the switches: one stream of draws)
                    if f_out:
                        t[vch] *= vamp[:, None]
                        sd[lp + nm + ".bias"][vch] = 1.5 * sgn
            q, k = once(lp + "self_attn.q_proj.weight"), once(lp + "self_attn.k_proj.weight")
            if q is not None and k is not None and f_attn:
                k.copy_(0.7 * q + math.sqrt(1 - 0.49) * k)
                q *= 2.0
            for nm in ("layer_norm1", "layer_norm2"):
                if not f_norm:
                    continue
                spread_gain(lp + nm + ".weight", vch, 0.7, 0.4, 0.1)
                b = once(lp + nm + ".bias")
                if b is not None:
                    b *= 5.0
    if dtype == torch.float32:
        return sd
    conv: Dict[int, torch.Tensor] = {}          # keep share_layers' aliasing: one converted tensor per distinct host tensor
    out = {}
    for k, v in sd.items():                     # (setdefault would evaluate v.to(dtype) for every alias: 32 x per shared tensor)
        if id(v) not in conv:
            conv[id(v)] = v.to(dtype)
        out[k] = conv[id(v)]
    return out


def dense_pe(gaussian: torch.Tensor, grid: int = 48) -> torch.Tensor:
    """PositionEmbeddingRandom.forward((48,48)) in the parameter dtype, as `prompt_encoder.get_dense_pe()` computes it
    (segment_anything/modeling/prompt_encoder.py:67-76,189-229).  Returns [grid*grid, 256] (channels-last)."""
    dt = gaussian.dtype
    ones = torch.ones((grid, grid), dtype=dt)
    y = (ones.cumsum(dim=0) - 0.5) / grid
    x = (ones.cumsum(dim=1) - 0.5) / grid
    coords = torch.stack([x, y], dim=-1)
    coords = 2 * coords - 1
    coords = coords @ gaussian
    coords = 2 * math.pi * coords
    pe = torch.cat([torch.sin(coords), torch.cos(coords)], dim=-1)  # [g, g, 256]
    return pe.reshape(grid * grid, 256).contiguous()


def _read_shards(d: str) -> Dict[str, torch.Tensor]:
    out: Dict[str, torch.Tensor] = {}
    names = sorted(os.listdir(d))
    st = [n for n in names if n.endswith(".safetensors")]
    if st:
        from safetensors.torch import load_file
        for n in st:
            out.update(load_file(os.path.join(d, n)))
        return out
    for n in names:
        if n.endswith(".bin") and n.startswith("pytorch_model"):
            out.update(torch.load(os.path.join(d, n), map_location="cpu", weights_only=True))
    if not out:
        raise FileNotFoundError(f"no weight shards in {d}")
    return out


def vqa_config_from_dir(vqa_dir: str, **overrides) -> VQAConfig:
    """VQAConfig from the checkpoint's config.json (LlavaSearchConfig = LlamaConfig + mm_* fields)."""
    import json
    c = json.load(open(os.path.join(vqa_dir, "config.json")))
    pt = c.get("mm_projector_type", "linear")
    if pt not in ("linear", "mlp2x_gelu"):
        raise ValueError(f"unsupported mm_projector_type {pt!r}")
    kw = dict(llm_hidden=c["hidden_size"], llm_heads=c["num_attention_heads"], llm_mlp=c["intermediate_size"],
              llm_layers=c["num_hidden_layers"], llm_vocab=c["vocab_size"], llm_rms_eps=c.get("rms_norm_eps", 1e-6),
              llm_rope_theta=c.get("rope_theta", 10000.0), projector_type=0 if pt == "linear" else 1,
              clip_select_layer=c.get("mm_vision_select_layer", -2))
    kw.update(overrides)
    return VQAConfig(**kw)


def load_vqa_checkpoint_dir(vqa_dir: str, clip_dir: str | None = None) -> Dict[str, torch.Tensor]:
    """HF `save_pretrained` directory of LlavaSearchLlamaForCausalLM (craigwu/seal_vqa_7b) -> engine key space.  The CLIP
    tower is taken from the checkpoint when it carries one (`model.vision_tower.vision_tower.*`), else from `clip_dir`
    (builder.py:137-140 loads it separately)."""
    sd: Dict[str, torch.Tensor] = {}
    vt = "model.vision_tower.vision_tower."
    for k, v in _read_shards(vqa_dir).items():
        if k.startswith(vt):
            sd[CLIP_PREFIX + k[len(vt):]] = v
        elif ".vision_tower." not in k:
            sd[k] = v
    if not any(k.startswith(CLIP_PREFIX) for k in sd):
        if clip_dir is None:
            raise FileNotFoundError("the checkpoint holds no CLIP tower; pass the openai/clip-vit-large-patch14 directory")
        for k, v in _read_shards(clip_dir).items():
            if k.startswith("vision_model."):
                sd[CLIP_PREFIX + k] = v
    return sd


def load_checkpoint_dir(vsm_dir: str, clip_dir: str) -> Dict[str, torch.Tensor]:
    """Reads a HF `save_pretrained` directory of the VSM (safetensors or .bin shards) plus the CLIP tower directory and
    returns one state dict in the engine's key space (CLIP keys prefixed with `clip.`)."""
    sd: Dict[str, torch.Tensor] = {}
    _read = _read_shards

    for k, v in _read(vsm_dir).items():
        if ".vision_tower." in k:
            continue
        sd[k] = v
    for k, v in _read(clip_dir).items():
        if k.startswith("vision_model."):
            sd[CLIP_PREFIX + k] = v
    return sd
