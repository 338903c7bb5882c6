"""CPU oracle for the VSM scoring path — TEST INFRASTRUCTURE ONLY.

A plain-torch (CPU) functional restatement of what the reference computes per crop:
  VSMForCausalLM.model_forward(inference=True)   VisualSearch/model/VSM.py:201-364
  (= the single-prefill form of VSMForCausalLM.inference, VSM.py:438-553)
Every function cites the reference lines it follows.  The arithmetic of CLIP / LLaMA / OWL-ViT lives in the reference's
third-party dependency `transformers==4.31.0` (requirements.txt:43), which is not vendored in the reference tree; those
blocks restate the published HF algorithms and are anchored on the reference's call sites.

Pinning: the reference has NO tests or golden vectors of its own.  This oracle is pinned against the
reference itself, imported in the build container under the shim recipe of oracle/ref_shim.py, on seeded random weights;
the resulting input/output vectors are committed under tests/golden/ together with oracle/gen_golden.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  The product path
(vstar_amd/) never does, and has no CPU fallback.

The computation dtype is the dtype of the state dict (fp32 = the parity oracle; bf16 = emulation of the reference's
bf16 run on torch CPU kernels).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


def _lin(sd: SD, key: str, x: torch.Tensor, bias: bool = True) -> torch.Tensor:
    return F.linear(x, sd[key + ".weight"], sd.get(key + ".bias") if bias else None)


def _ln(sd: SD, key: str, x: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[key + ".weight"], sd[key + ".bias"], eps)


def quick_gelu(x: torch.Tensor) -> torch.Tensor:
    # HF QuickGELUActivation (CLIP / OWL-ViT hidden_act="quick_gelu")
    return x * torch.sigmoid(1.702 * x)


# ------------------------------------------------------------------------------------------------------------
# ViT towers (HF CLIPVisionTransformer / OwlViTVisionTransformer; call sites clip_encoder.py:53-57, owlvit.py:121-126)
# ------------------------------------------------------------------------------------------------------------
def vit_tower(sd: SD, prefix: str, preln: str, pix: torch.Tensor, heads: int, n_blocks: int) -> torch.Tensor:
    """Returns the hidden state after `n_blocks` pre-LN encoder blocks, [B, 1+P, C]."""
    w = sd[prefix + "embeddings.patch_embedding.weight"]
    ps = w.shape[-1]
    x = F.conv2d(pix.to(w.dtype), w, None, stride=ps)             # [B, C, g, g]
    x = x.flatten(2).transpose(1, 2)                              # [B, P, C]
    cls = sd[prefix + "embeddings.class_embedding"].expand(x.shape[0], 1, -1)
    x = torch.cat([cls, x], dim=1) + sd[prefix + "embeddings.position_embedding.weight"].unsqueeze(0)
    x = _ln(sd, prefix + preln, x)
    B, N, C = x.shape
    hd = C // heads
    for i in range(n_blocks):
        lp = f"{prefix}encoder.layers.{i}."
        h = _ln(sd, lp + "layer_norm1", x)
        q = _lin(sd, lp + "self_attn.q_proj", h) * (hd ** -0.5)
        k = _lin(sd, lp + "self_attn.k_proj", h)
        v = _lin(sd, lp + "self_attn.v_proj", h)
        q, k, v = (t.view(B, N, heads, hd).transpose(1, 2) for t in (q, k, v))
        att = torch.softmax(q @ k.transpose(-1, -2), dim=-1) @ v
        att = att.transpose(1, 2).reshape(B, N, C)
        x = x + _lin(sd, lp + "self_attn.out_proj", att)
        h = _ln(sd, lp + "layer_norm2", x)
        x = x + _lin(sd, lp + "mlp.fc2", quick_gelu(_lin(sd, lp + "mlp.fc1", h)))
    return x


def clip_features(sd: SD, pix: torch.Tensor, heads: int, layers: int, select_layer: int = -2) -> torch.Tensor:
    """CLIPVisionTower.forward + feature_select('patch'): hidden_states[select_layer][:, 1:] (clip_encoder.py:31-60)."""
    n_blocks = layers + 1 + select_layer
    return vit_tower(sd, "clip.vision_model.", "pre_layrnorm", pix, heads, n_blocks)[:, 1:]


def encode_images(sd: SD, pix: torch.Tensor, heads: int, layers: int, select_layer: int = -2) -> torch.Tensor:
    """LlavaMetaForCausalLM.encode_images: vision tower -> mm_projector (llava_arch.py:93-96)."""
    return _lin(sd, "model.mm_projector", clip_features(sd, pix, heads, layers, select_layer))


def splice(sd: SD, input_ids: torch.Tensor, image_features: torch.Tensor) -> torch.Tensor:
    """prepare_inputs_labels_for_multimodal, mm_use_im_start_end branch (llava_arch.py:185-208,235-247,328-345):
    the single -200 is replaced by the P projected rows; every other id is embedded."""
    emb = sd["model.embed_tokens.weight"]
    rows = []
    for b in range(input_ids.shape[0]):
        ids = input_ids[b]
        pos = torch.where(ids == -200)[0]
        assert pos.numel() == 1, "exactly one image token per sample"
        p = int(pos[0])
        rows.append(torch.cat([emb[ids[:p]], image_features[b].to(emb.dtype), emb[ids[p + 1:]]], dim=0))
    return torch.stack(rows, dim=0)


# ------------------------------------------------------------------------------------------------------------
# LLaMA prefill (HF LlamaModel 4.31 semantics; call site llava_llama.py:93-102, returns the final-normed hidden state
# as `hidden_states` in eval mode, llava_llama.py:124-133)
# ------------------------------------------------------------------------------------------------------------
def rms_norm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    dt = x.dtype
    xf = x.float()
    xf = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
    return w * xf.to(dt)


def rope_tables(S: int, hd: int, theta: float, dtype: torch.dtype, device=None) -> Tuple[torch.Tensor, torch.Tensor]:
    inv = 1.0 / (theta ** (torch.arange(0, hd, 2, dtype=torch.float32, device=device) / hd))
    f = torch.outer(torch.arange(S, dtype=torch.float32, device=device), inv)
    emb = torch.cat([f, f], dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def rotate_half(x: torch.Tensor) -> torch.Tensor:
    h = x.shape[-1] // 2
    return torch.cat([-x[..., h:], x[..., :h]], dim=-1)


def fp8_fake_quant(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Per-row symmetric OCP e4m3 quantisation as the engine's W8A8 mode does it (vstar_amd/csrc/quant.hip; BASELINE config 5
    — the reference has no fp8 path): scale = absmax / 448, value * (1 / scale) rounded to nearest-even e4m3.  Returns the
    decoded codes (fp32) and the scales."""
    s = x.float().abs().amax(dim=-1, keepdim=True) / 448.0
    s = torch.where(s > 0, s, torch.ones_like(s))
    return (x.float() * (1.0 / s)).to(torch.float8_e4m3fn).float(), s


def linear_w8a8(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """x @ w.T with per-token activation scales and per-output-channel weight scales, fp32 accumulation."""
    xq, sx = fp8_fake_quant(x)
    wq, sw = fp8_fake_quant(w)
    return ((xq @ wq.T) * sx * sw.transpose(-1, -2)).to(x.dtype)


def mx_fake_quant(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Block-scaled (MX style) OCP e4m3 quantisation as the engine's W8A8 mode does it for the inputs of o_proj / down_proj since round 6
    (vstar_amd/csrc/mx.hpp; the reference has no fp8 path): blocks of 32 consecutive values along the last dim, ONE E8M0 byte e per
    block = the smallest power of two 2^(e - 127) with amax / 2^(e - 127) <= 448 — in integer arithmetic on the fp32 bits of amax
    (= 1.f x 2^E): e = E + 127 - 8, plus one when 1.f > 1.75 — and value * 2^(127 - e) rounded to nearest-even e4m3.  Returns the
    DECODED values (fp32, code x block scale) and the E8M0 bytes [..., n / 32]."""
    xf = x.float()
    blk = xf.reshape(*xf.shape[:-1], xf.shape[-1] // 32, 32)
    amax = blk.abs().amax(dim=-1, keepdim=True).contiguous()
    bits = amax.view(torch.int32)
    e = ((bits >> 23) - 8 + ((bits & 0x7FFFFF) > 0x600000).to(torch.int32)).clamp_min(0)
    inv = ((254 - e) << 23).view(torch.float32)                      # 2^(127 - e), exact
    codes = (blk * inv).to(torch.float8_e4m3fn).float()
    scale = torch.ldexp(torch.ones_like(amax), e - 127)
    return (codes * scale).reshape(xf.shape), e.squeeze(-1).to(torch.uint8)


def linear_w8a8_mx(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """x @ w.T with block-scaled activations (mx_fake_quant) and per-output-channel weight scales, fp32 accumulation."""
    xd, _ = mx_fake_quant(x)
    wq, sw = fp8_fake_quant(w)
    return ((xd @ wq.T) * sw.transpose(-1, -2)).to(x.dtype)


def linear_w8a8_mx_folded(x: torch.Tensor, w: torch.Tensor, g: torch.Tensor, eps: float) -> torch.Tensor:
    """Linear(RMSNorm_g(x)) as the fully block-scaled chain runs it (engine level 2): the RAW rows x block-scaled, the norm weight folded
    into the columns of W (rounded to the storage type, then per-output-channel e4m3), 1 / rms(x) applied to the accumulators."""
    xd, _ = mx_fake_quant(x)
    wq, sw = fp8_fake_quant((w.float() * g.float()).to(w.dtype))
    rstd = torch.rsqrt(x.float().pow(2).mean(-1, keepdim=True) + eps)
    return ((xd @ wq.T) * rstd * sw.transpose(-1, -2)).to(x.dtype)


def llama_prefill(sd: SD, x: torch.Tensor, heads: int, layers: int, eps: float, theta: float,
                  final_norm: bool = True, w8a8: bool = False, mx: int = 0) -> torch.Tensor:
    """w8a8: the four big linears of every block on fake-quantised operands, as the engine's config-5 mode runs them —
    except o_proj / MLP of the LAST block, which the engine evaluates on the few needed rows with the 16-bit weights.
    mx (with w8a8; VstarEngine.w8a8_mx_active()): 0 = per-token activation scales everywhere; 1 = the inputs of o_proj and down_proj
    block-scaled; 2 = also the inputs of q|k|v and gate|up (raw residual stream block-scaled, RMSNorm folded: linear_w8a8_mx_folded)."""
    B, S, H = x.shape
    mx = int(mx)
    lin8 = (lambda key, t: (linear_w8a8_mx if mx and key.endswith(("o_proj", "down_proj")) else linear_w8a8)(t, sd[key + ".weight"])) if w8a8 else None
    hd = H // heads
    cos, sin = rope_tables(S, hd, theta, x.dtype, x.device)
    mask = torch.full((S, S), float("-inf"), device=x.device).triu(1)
    for i in range(layers):
        lp = f"model.layers.{i}."
        h = rms_norm(x, sd[lp + "input_layernorm.weight"], eps)
        proj = (lambda key, t, last_ok=True: lin8(key, t)) if w8a8 else (lambda key, t, last_ok=True: _lin(sd, key, t, False))
        post = proj if (not w8a8 or i + 1 < layers) else (lambda key, t: _lin(sd, key, t, False))
        if w8a8 and mx >= 2:      # q|k|v / gate|up straight from the raw rows (their norm folded); `h` is then unused by them
            x_in = x
            proj = lambda key, t, g=sd[lp + "input_layernorm.weight"], x_in=x_in: linear_w8a8_mx_folded(x_in, sd[key + ".weight"], g, eps)  # noqa: E731
        q = proj(lp + "self_attn.q_proj", h).view(B, S, heads, hd).transpose(1, 2)
        k = proj(lp + "self_attn.k_proj", h).view(B, S, heads, hd).transpose(1, 2)
        v = proj(lp + "self_attn.v_proj", h).view(B, S, heads, hd).transpose(1, 2)
        q = q * cos + rotate_half(q) * sin
        k = k * cos + rotate_half(k) * sin
        w = (q @ k.transpose(-1, -2)) / math.sqrt(hd) + mask.to(q.dtype)
        w = torch.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)
        att = (w @ v).transpose(1, 2).reshape(B, S, H)
        x = x + post(lp + "self_attn.o_proj", att)
        h = rms_norm(x, sd[lp + "post_attention_layernorm.weight"], eps)
        gu = post
        if w8a8 and mx >= 2 and i + 1 < layers:
            gu = lambda key, t, g=sd[lp + "post_attention_layernorm.weight"], x_in=x: linear_w8a8_mx_folded(x_in, sd[key + ".weight"], g, eps)  # noqa: E731
        x = x + post(lp + "mlp.down_proj", F.silu(gu(lp + "mlp.gate_proj", h)) * gu(lp + "mlp.up_proj", h))
    return rms_norm(x, sd["model.norm.weight"], eps) if final_norm else x


def text_hidden_fcs(sd: SD, branch: str, h: torch.Tensor) -> torch.Tensor:
    """text_hidden_fcs_{det,seg}: Linear -> ReLU -> Linear -> Dropout(0) (VSM.py:120-140)."""
    p = f"model.text_hidden_fcs_{branch}.0."
    return _lin(sd, p + "2", F.relu(_lin(sd, p + "0", h)))


# ------------------------------------------------------------------------------------------------------------
# OWL-ViT wrapper (VisualSearch/model/owlvit/owlvit.py)
# ------------------------------------------------------------------------------------------------------------
def owl_visual_embs(sd: SD, pix: torch.Tensor, heads: int, layers: int) -> torch.Tensor:
    """OwlViT.get_visual_embs (owlvit.py:121-148) -> [B, g, g, C]."""
    pre = "model.owlvit.vision_model."
    x = vit_tower(sd, pre, "pre_layernorm", pix, heads, layers)
    x = _ln(sd, pre + "post_layernorm", x)
    x = x[:, 1:, :] * x[:, :1, :]
    x = _ln(sd, "model.owlvit.layer_norm", x)
    g = int(math.isqrt(x.shape[1]))
    return x.reshape(x.shape[0], g, g, x.shape[-1])


def owl_box_bias(g: int) -> torch.Tensor:
    """compute_box_bias (owlvit.py:42-77)."""
    coords = np.stack(np.meshgrid(np.arange(1, g + 1), np.arange(1, g + 1)), axis=-1).astype(np.float32)
    coords /= np.array([g, g], np.float32)
    coords = torch.from_numpy(coords.reshape(g * g, 2)).clip(0.0, 1.0)
    cb = torch.log(coords + 1e-4) - torch.log1p(-coords + 1e-4)
    size = torch.full_like(cb, 1.0 / g)
    sb = torch.log(size + 1e-4) - torch.log1p(-size + 1e-4)
    return torch.cat([cb, sb], dim=-1)


def owl_heads(sd: SD, feature_map: torch.Tensor, query: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """OwlViT.forward (owlvit.py:150-170) with HF OwlViTClassPredictionHead / OwlViTBoxPredictionHead.
    feature_map [B,g,g,C], query [B,1,Q] -> pred_logits [B,g*g,1], pred_boxes [B,g*g,4]."""
    B, g, _, C = feature_map.shape
    feats = feature_map.reshape(B, g * g, C)
    ch = "model.owlvit.class_head."
    e = _lin(sd, ch + "dense0", feats)
    e = e / (torch.linalg.norm(e, dim=-1, keepdim=True) + 1e-6)
    q = query / (torch.linalg.norm(query, dim=-1, keepdim=True) + 1e-6)
    logits = torch.einsum("...pd,...qd->...pq", e, q)
    shift = _lin(sd, ch + "logit_shift", feats)
    scale = F.elu(_lin(sd, ch + "logit_scale", feats)) + 1
    logits = (logits + shift) * scale
    bh = "model.owlvit.box_head."
    b = _lin(sd, bh + "dense2", F.gelu(_lin(sd, bh + "dense1", F.gelu(_lin(sd, bh + "dense0", feats)))))
    b += owl_box_bias(g).to(b.device)
    return logits, torch.sigmoid(b)


# ------------------------------------------------------------------------------------------------------------
# SAM-style mask head (segment_anything/modeling/{prompt_encoder,mask_decoder,transformer,common}.py)
# ------------------------------------------------------------------------------------------------------------
def dense_pe(sd: SD, grid: int = 48) -> torch.Tensor:
    """PromptEncoder.get_dense_pe -> [1, 256, g, g] (prompt_encoder.py:67-76,216-229)."""
    gm = sd["model.prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"]
    ones = torch.ones((grid, grid), dtype=gm.dtype, device=gm.device)
    y = (ones.cumsum(dim=0) - 0.5) / grid
    x = (ones.cumsum(dim=1) - 0.5) / grid
    c = 2 * torch.stack([x, y], dim=-1) - 1
    c = 2 * np.pi * (c @ gm)
    return torch.cat([torch.sin(c), torch.cos(c)], dim=-1).permute(2, 0, 1).unsqueeze(0)


def _sam_attention(sd: SD, p: str, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int = 8) -> torch.Tensor:
    """Attention.forward (transformer.py:220-242)."""
    q, k, v = _lin(sd, p + "q_proj", q), _lin(sd, p + "k_proj", k), _lin(sd, p + "v_proj", v)

    def sep(t):
        b, n, c = t.shape
        return t.reshape(b, n, heads, c // heads).transpose(1, 2)

    q, k, v = sep(q), sep(k), sep(v)
    a = torch.softmax((q @ k.permute(0, 1, 3, 2)) / math.sqrt(q.shape[-1]), dim=-1) @ v
    b, h, n, c = a.shape
    return _lin(sd, p + "out_proj", a.transpose(1, 2).reshape(b, n, h * c))


def two_way_transformer(sd: SD, src: torch.Tensor, pos: torch.Tensor, tokens: torch.Tensor):
    """TwoWayTransformer.forward, depth 2 (transformer.py:62-107,151-182)."""
    p = "model.mask_decoder.transformer."
    keys = src.flatten(2).permute(0, 2, 1)
    key_pe = pos.flatten(2).permute(0, 2, 1)
    queries, query_pe = tokens, tokens
    for i in range(2):
        lp = f"{p}layers.{i}."
        if i == 0:
            queries = _sam_attention(sd, lp + "self_attn.", queries, queries, queries)
        else:
            q = queries + query_pe
            queries = queries + _sam_attention(sd, lp + "self_attn.", q, q, queries)
        queries = _ln(sd, lp + "norm1", queries)
        q, k = queries + query_pe, keys + key_pe
        queries = _ln(sd, lp + "norm2", queries + _sam_attention(sd, lp + "cross_attn_token_to_image.", q, k, keys))
        queries = _ln(sd, lp + "norm3", queries + _lin(sd, lp + "mlp.lin2", F.relu(_lin(sd, lp + "mlp.lin1", queries))))
        q, k = queries + query_pe, keys + key_pe
        keys = _ln(sd, lp + "norm4", keys + _sam_attention(sd, lp + "cross_attn_image_to_token.", k, q, queries))
    q, k = queries + query_pe, keys + key_pe
    queries = _ln(sd, p + "norm_final_attn", queries + _sam_attention(sd, p + "final_attn_token_to_image.", q, k, keys))
    return queries, keys


def _layer_norm_2d(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return w[:, None, None] * x + b[:, None, None]


def _upsample_conv(sd: SD, key: str, x: torch.Tensor) -> torch.Tensor:
    """mask_decoder.Upsample (mask_decoder.py:15-27)."""
    x = F.interpolate(x.float(), scale_factor=2.0, mode="bilinear").to(x.dtype)
    return F.conv2d(x, sd[key + ".conv.weight"], sd[key + ".conv.bias"], padding=1)


def sam_mask_head(sd: SD, feature_map: torch.Tensor, seg_embed: torch.Tensor, taps: Optional[dict] = None) -> torch.Tensor:
    """visual_projection + PromptEncoder(text_embeds) + MaskDecoder(multimask_output=False) for ONE [LOC] per crop
    (VSM.py:515-533; prompt_encoder.py:140-186; mask_decoder.py:96-186).  feature_map [B,g,g,C], seg_embed [B,256]
    -> low-res mask logits [B,1,4g,4g]."""
    md = "model.mask_decoder."
    B, g = feature_map.shape[0], feature_map.shape[1]
    img = F.linear(feature_map, sd["model.visual_projection.weight"]).permute(0, 3, 1, 2)        # [B,256,g,g]
    sparse = seg_embed.unsqueeze(1)                                                               # [B,1,256]
    dense = sd["model.prompt_encoder.no_mask_embed.weight"].reshape(1, -1, 1, 1).expand(1, -1, g, g)
    pe = dense_pe(sd, g).to(img.dtype)
    out = []
    for i in range(B):  # the reference loops over crops with batch 1 (VSM.py:312-330)
        tokens = torch.cat([sd[md + "iou_token.weight"], sd[md + "mask_tokens.weight"]], dim=0).unsqueeze(0)
        tokens = torch.cat([tokens, sparse[i:i + 1].to(tokens.dtype)], dim=1)                     # [1,6,256]
        src = img[i:i + 1] + dense
        hs, keys = two_way_transformer(sd, src, pe, tokens)
        mask_tokens_out = hs[:, 1:5, :]
        x = keys.transpose(1, 2).view(1, 256, g, g)
        x = _upsample_conv(sd, md + "output_upscaling.0", x)
        if taps is not None:       # stage taps (channels-last, like the engine's buffers) for the mask-head error budget
            taps.setdefault("sam_src", []).append(src.permute(0, 2, 3, 1).reshape(-1, 256) - dense.permute(0, 2, 3, 1).reshape(-1, 256))
            taps.setdefault("sam_tokens", []).append(hs[0])
            taps.setdefault("sam_keys", []).append(keys[0])
            taps.setdefault("sam_c1", []).append(x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]))
        x = F.gelu(_layer_norm_2d(x, sd[md + "output_upscaling.1.weight"], sd[md + "output_upscaling.1.bias"]))
        if taps is not None:
            taps.setdefault("sam_c1n", []).append(x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]))
        x = F.gelu(_upsample_conv(sd, md + "output_upscaling.3", x))
        t = mask_tokens_out[:, 0, :]
        for j in range(3):
            t = _lin(sd, md + f"output_hypernetworks_mlps.0.layers.{j}", t)
            if j < 2:
                t = F.relu(t)
        if taps is not None:
            taps.setdefault("sam_c2", []).append(x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]))
            taps.setdefault("sam_hyper", []).append(t[0])
        b, c, h, w = x.shape
        out.append((t.unsqueeze(1) @ x.view(b, c, h * w)).view(b, 1, h, w))
    return torch.cat(out, dim=0)


def upsample_mask(low_res: torch.Tensor, size: Tuple[int, int]) -> torch.Tensor:
    """F.interpolate(low_res.float(), (h, w), bilinear, align_corners=False) then clamp(min=0)
    (VSM.py:534-537; visual_search.py:223-224)."""
    return torch.clamp(F.interpolate(low_res.float(), size, mode="bilinear", align_corners=False), min=0)


# ------------------------------------------------------------------------------------------------------------
# The whole path
# ------------------------------------------------------------------------------------------------------------
def vsm_forward(sd: SD, cfg, images_clip: torch.Tensor, images: Optional[torch.Tensor], input_ids: torch.Tensor,
                loc_token_idx: int, verify_pos: Optional[torch.Tensor] = None, w8a8_mx: int = 0) -> Dict[str, torch.Tensor]:
    """model_forward(inference=True) batched over independent crops (each crop = one reference call with batch 1).
    `cfg` is a vstar_amd.config.VSMConfig (only its integer fields are read).  w8a8_mx: with cfg.llm_w8a8, the block-scaled scheme for
    the inputs of o_proj / down_proj (what the engine ran: VstarEngine.w8a8_mx_active())."""
    dt = sd["model.norm.weight"].dtype
    P = (cfg.clip_image_size // cfg.clip_patch) ** 2
    feats = clip_features(sd, images_clip.to(dt), cfg.clip_heads, cfg.clip_layers, cfg.clip_select_layer)
    proj = _lin(sd, "model.mm_projector", feats)
    x = splice(sd, input_ids, proj)
    hidden = llama_prefill(sd, x, cfg.llm_heads, cfg.llm_layers, cfg.llm_rms_eps, cfg.llm_rope_theta,
                           w8a8=bool(getattr(cfg, "llm_w8a8", 0)), mx=w8a8_mx)
    # loc_token_mask = (input_ids[:,1:] == loc) shifted right by P-1 (VSM.py:224-235,465-473): selects the hidden
    # state at spliced index idx([LOC]) - 1 + (P - 1)
    B = input_ids.shape[0]
    loc_pos = []
    for b in range(B):
        w = torch.where(input_ids[b] == loc_token_idx)[0]
        assert w.numel() == 1, "one [LOC] per crop"
        loc_pos.append(int(w[0]) - 1 + (P - 1))
    loc_pos_t = torch.tensor(loc_pos, device=hidden.device)
    h_loc = hidden[torch.arange(B, device=hidden.device), loc_pos_t]
    out = {"clip_features": feats, "projector": proj, "llm_hidden_loc": h_loc, "loc_pos": loc_pos_t,
           "embed_det": text_hidden_fcs(sd, "det", h_loc), "embed_seg": text_hidden_fcs(sd, "seg", h_loc)}
    if verify_pos is not None:
        hv = hidden[torch.arange(B, device=hidden.device).unsqueeze(1), torch.as_tensor(verify_pos, device=hidden.device)]
        out["tf_logits"] = F.linear(hv, sd["lm_head.weight"]).float()        # [B, V, vocab]: lets a test judge arg-max margins
        out["tf_argmax"] = out["tf_logits"].argmax(-1)
    if images is not None:
        fmap = owl_visual_embs(sd, images.to(dt), cfg.owl_heads, cfg.owl_layers)
        out["owl_feats"] = fmap
        logits, boxes = owl_heads(sd, fmap, out["embed_det"].unsqueeze(1))
        out["pred_logits"], out["pred_boxes"] = logits, boxes
        taps: dict = {}
        out["low_res_masks"] = sam_mask_head(sd, fmap, out["embed_seg"], taps)
        out["sam_taps"] = {k: torch.stack(v) for k, v in taps.items()}
    return out


def greedy_next_logits(sd: SD, cfg, images_clip: torch.Tensor, input_ids: torch.Tensor) -> torch.Tensor:
    """One step of the reference's no-cache greedy decoding (VSM.py:451-458 with use_cache=False, VSM.py:151): a full
    forward over the sequence so far; returns lm_head logits of the LAST position, [B, vocab] fp32."""
    dt = sd["model.norm.weight"].dtype
    feats = clip_features(sd, images_clip.to(dt), cfg.clip_heads, cfg.clip_layers, cfg.clip_select_layer)
    x = splice(sd, input_ids, _lin(sd, "model.mm_projector", feats))
    hidden = llama_prefill(sd, x, cfg.llm_heads, cfg.llm_layers, cfg.llm_rms_eps, cfg.llm_rope_theta)
    return F.linear(hidden[:, -1], sd["lm_head.weight"]).float()
