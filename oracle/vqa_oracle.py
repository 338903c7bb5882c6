"""CPU oracle for the VQA-LLM path — TEST INFRASTRUCTURE ONLY.

A plain-torch (CPU) functional restatement of `LlavaSearchLlamaForCausalLM` as the V* evaluation drives it:
  encode_images / project_features ......... LLaVA/llava/model/llava_search_arch.py:84-94
  PerceiverResampler ....................... LLaVA/llava/model/multimodal_projector/perceiver.py:25-121
  projector builder ........................ LLaVA/llava/model/multimodal_projector/builder.py:33-68
  <image>/<object> splice .................. LLaVA/llava/model/llava_search_arch.py:96-266 (batch of one, no labels)
  forward / KV cache ....................... LLaVA/llava/model/language_model/llava_search_llama.py:56-113
  free-form greedy decode .................. vstar_bench_eval.py:78-113   (temperature 0 => greedy)
  multiple-choice option scoring ........... vstar_bench_eval.py:115-165
The LLaMA / CLIP arithmetic lives in `transformers==4.31.0` (requirements.txt:43), not vendored in the reference tree;
those blocks restate the published HF algorithms and are anchored on the call sites above.

Pinning: the reference holds no tests or golden vectors for this path.  The oracle is pinned against the reference's own
modules imported under oracle/vqa_ref_shim.py (build container only), seeded random weights, tiny widths with the real
topology; the vectors are committed under tests/golden/vqa_*.npz with the generating script oracle/gen_vqa_golden.py.

Only tests/, __graft_entry__.smoke() and bench tooling's cpu legs may import this module; vstar_amd/ never does.
The computation dtype is the dtype of the state dict (fp32 = parity oracle; fp16 = emulation of the reference's run).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from .vsm_oracle import clip_features, rms_norm, rope_tables, rotate_half

SD = Dict[str, torch.Tensor]
IMAGE_TOKEN_INDEX = -200    # LLaVA/llava/constants.py:9
OBJECT_TOKEN_INDEX = -300   # LLaVA/llava/constants.py:10
PO = "model.mm_projector_object."


def _ln(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    return F.layer_norm(x, (x.shape[-1],), w, b, 1e-5)


# ------------------------------------------------------------------------------------------------------------
# PerceiverResampler (perceiver.py:25-121): x [b, n, d] -> [b, num_latents, d]
# ------------------------------------------------------------------------------------------------------------
def perceiver_resampler(sd: SD, x: torch.Tensor, depth: int, heads: int, dim_head: int) -> torch.Tensor:
    p = PO + "1."
    b, n, d = x.shape
    x = x + sd[p + "media_pos_emb"][:1].reshape(1, 1, d)                       # perceiver.py:106-107 (times = 1)
    lat = sd[p + "latents"].unsqueeze(0).expand(b, -1, -1)                     # perceiver.py:109
    scale = dim_head ** -0.5
    for i in range(depth):
        a = f"{p}layers.{i}.0."
        f = f"{p}layers.{i}.1."
        xm = _ln(x, sd[a + "norm_media.weight"], sd[a + "norm_media.bias"])            # perceiver.py:52-53
        lt = _ln(lat, sd[a + "norm_latents.weight"], sd[a + "norm_latents.bias"])
        q = F.linear(lt, sd[a + "to_q.weight"])
        kv = F.linear(torch.cat([xm, lt], dim=-2), sd[a + "to_kv.weight"])            # perceiver.py:60-61
        k, v = kv.chunk(2, dim=-1)
        sep = lambda t: t.reshape(b, t.shape[1], heads, dim_head).transpose(1, 2)      # 'b n (h d) -> b h n d'
        q, k, v = sep(q) * scale, sep(k), sep(v)                                       # perceiver.py:65
        sim = q @ k.transpose(-1, -2)
        sim = sim - sim.amax(dim=-1, keepdim=True)                                     # perceiver.py:71
        attn = sim.softmax(dim=-1)
        out = (attn @ v).transpose(1, 2).reshape(b, -1, heads * dim_head)
        lat = F.linear(out, sd[a + "to_out.weight"]) + lat                             # perceiver.py:112
        h = _ln(lat, sd[f + "0.weight"], sd[f + "0.bias"])                             # FeedForward, perceiver.py:17-23
        h = F.linear(F.gelu(F.linear(h, sd[f + "1.weight"])), sd[f + "3.weight"])
        lat = h + lat                                                                  # perceiver.py:113
    return _ln(lat, sd[p + "norm.weight"], sd[p + "norm.bias"])                        # perceiver.py:115


def project_long(sd: SD, feats: torch.Tensor, projector_type: int) -> torch.Tensor:
    """mm_projector (builder.py:39-49): linear or mlp2x_gelu."""
    if projector_type == 0:
        return F.linear(feats, sd["model.mm_projector.weight"], sd["model.mm_projector.bias"])
    h = F.gelu(F.linear(feats, sd["model.mm_projector.0.weight"], sd["model.mm_projector.0.bias"]))
    return F.linear(h, sd["model.mm_projector.2.weight"], sd["model.mm_projector.2.bias"])


def project_short(sd: SD, feats: torch.Tensor, cfg) -> torch.Tensor:
    """mm_projector_object = Sequential(LayerNorm, PerceiverResampler, Linear) (builder.py:54-66)."""
    x = _ln(feats, sd[PO + "0.weight"], sd[PO + "0.bias"])
    x = perceiver_resampler(sd, x, cfg.pcv_depth, cfg.pcv_heads, cfg.pcv_dim_head)
    return F.linear(x, sd[PO + "2.weight"], sd[PO + "2.bias"])


def encode_images(sd: SD, cfg, pix: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """encode_images == project_features (llava_search_arch.py:84-94): pix [n,3,I,I] -> long [n,P,H], short [n,L,H]."""
    feats = clip_features(sd, pix, cfg.clip_heads, cfg.clip_layers, cfg.clip_select_layer)   # patch tokens, CLS dropped
    return project_long(sd, feats, cfg.projector_type), project_short(sd, feats, cfg)


# ------------------------------------------------------------------------------------------------------------
# prepare_inputs_labels_for_multimodal (llava_search_arch.py:96-266), one sample, inference (no labels, mask of ones)
# ------------------------------------------------------------------------------------------------------------
def splice(sd: SD, input_ids: Sequence[int], image_long: torch.Tensor, image_short: torch.Tensor,
           obj_long: Optional[torch.Tensor], obj_short: Optional[torch.Tensor], images_long: Optional[Sequence[bool]],
           objects_long: Optional[Sequence[bool]]) -> torch.Tensor:
    """image_long/short: [n_img, P|L, H]; obj_long/short: [n_obj, P|L, H].  Returns inputs_embeds [S, H]."""
    table = sd["model.embed_tokens.weight"]
    ids = list(input_ids)
    parts: List[torch.Tensor] = []
    cur: List[int] = []
    i_img = i_obj = 0

    def flush():
        if cur:
            parts.append(table[torch.tensor(cur, dtype=torch.long)])
            cur.clear()

    # the reference consumes every <image> first, then every <object> of the remaining ids (:136-200); in the prompts
    # of this pipeline <image> always precedes the <object> tokens, so one left-to-right pass is the same thing
    assert all(ids.index(IMAGE_TOKEN_INDEX) < k for k, t in enumerate(ids) if t == OBJECT_TOKEN_INDEX) or \
        IMAGE_TOKEN_INDEX not in ids or OBJECT_TOKEN_INDEX not in ids
    for t in ids:
        if t == IMAGE_TOKEN_INDEX:
            flush()
            use_long = images_long is None or images_long[i_img]                        # :137-140
            parts.append(image_long[i_img] if use_long else image_short[i_img])
            i_img += 1
        elif t == OBJECT_TOKEN_INDEX:
            flush()
            use_long = not (objects_long is None or not objects_long[i_obj])            # :176-179
            parts.append(obj_long[i_obj] if use_long else obj_short[i_obj])
            i_obj += 1
        else:
            cur.append(t)
    flush()
    return torch.cat(parts, dim=0)


# ------------------------------------------------------------------------------------------------------------
# LLaMA with a KV cache (HF LlamaModel 4.31; llava_search_llama.py:79-93)
# ------------------------------------------------------------------------------------------------------------
Past = List[Tuple[torch.Tensor, torch.Tensor]]


def llama_forward(sd: SD, cfg, x: torch.Tensor, past: Optional[Past] = None) -> Tuple[torch.Tensor, Past]:
    """x: inputs_embeds [T, H] (one sequence).  Returns (logits [T, vocab], new past)."""
    T, H = x.shape
    heads, hd = cfg.llm_heads, H // cfg.llm_heads
    P0 = 0 if past is None else past[0][0].shape[1]
    cos, sin = rope_tables(P0 + T, hd, cfg.llm_rope_theta, x.dtype)
    cos, sin = cos[P0:], sin[P0:]
    mask = torch.full((T, P0 + T), float("-inf")).triu(P0 + 1)
    new_past: Past = []
    for i in range(cfg.llm_layers):
        lp = f"model.layers.{i}."
        h = rms_norm(x, sd[lp + "input_layernorm.weight"], cfg.llm_rms_eps)
        q = F.linear(h, sd[lp + "self_attn.q_proj.weight"]).view(T, heads, hd).transpose(0, 1)
        k = F.linear(h, sd[lp + "self_attn.k_proj.weight"]).view(T, heads, hd).transpose(0, 1)
        v = F.linear(h, sd[lp + "self_attn.v_proj.weight"]).view(T, heads, hd).transpose(0, 1)
        q = q * cos + rotate_half(q) * sin
        k = k * cos + rotate_half(k) * sin
        if past is not None:
            k = torch.cat([past[i][0], k], dim=1)
            v = torch.cat([past[i][1], v], dim=1)
        new_past.append((k, v))
        w = (q @ k.transpose(-1, -2)) / math.sqrt(hd) + mask.to(q.dtype)
        w = torch.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)
        att = (w @ v).transpose(0, 1).reshape(T, H)
        x = x + F.linear(att, sd[lp + "self_attn.o_proj.weight"])
        h = rms_norm(x, sd[lp + "post_attention_layernorm.weight"], cfg.llm_rms_eps)
        x = x + F.linear(F.silu(F.linear(h, sd[lp + "mlp.gate_proj.weight"])) * F.linear(h, sd[lp + "mlp.up_proj.weight"]),
                         sd[lp + "mlp.down_proj.weight"])
    x = rms_norm(x, sd["model.norm.weight"], cfg.llm_rms_eps)
    return F.linear(x, sd["lm_head.weight"]), new_past


def option_loss(question_logits_last: torch.Tensor, option_logits: torch.Tensor, option_ids: Sequence[int]) -> torch.Tensor:
    """vstar_bench_eval.py:153-159: logits = cat(question_logits[-1:], option_logits[:-1]); CrossEntropyLoss (mean)."""
    logits = torch.cat([question_logits_last.reshape(1, -1), option_logits[:-1]], dim=0)
    return F.cross_entropy(logits.float(), torch.tensor(list(option_ids), dtype=torch.long)).to(logits.dtype)


def multiple_choice(sd: SD, cfg, question_embeds: torch.Tensor, option_ids: Sequence[Sequence[int]]):
    """multiple_choices_inference (vstar_bench_eval.py:115-165) after tokenisation: returns (losses, argmin)."""
    table = sd["model.embed_tokens.weight"]
    q_logits, past = llama_forward(sd, cfg, question_embeds)
    losses = []
    for ids in option_ids:
        o_logits, _ = llama_forward(sd, cfg, table[torch.tensor(list(ids), dtype=torch.long)], past)
        losses.append(option_loss(q_logits[-1], o_logits, ids))
    losses = torch.stack(losses)
    return losses, int(losses.argmin())


def greedy_generate(sd: SD, cfg, prompt_embeds: torch.Tensor, max_new_tokens: int, eos_id: int = 2) -> List[int]:
    """model.generate(do_sample=False, use_cache=True) (vstar_bench_eval.py:90-103), without the keyword stopping rule."""
    table = sd["model.embed_tokens.weight"]
    logits, past = llama_forward(sd, cfg, prompt_embeds)
    out: List[int] = []
    for _ in range(max_new_tokens):
        tok = int(logits[-1].float().argmax())
        out.append(tok)
        if tok == eos_id:
            break
        logits, past = llama_forward(sd, cfg, table[torch.tensor([tok])], past)
    return out
