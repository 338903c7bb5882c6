#!/usr/bin/env python
"""Error budget of the SAM-style mask head, stage by stage: engine taps vs the fp32 oracle evaluated on the
SAME bf16-rounded weights and inputs, so every number is pure arithmetic noise.  Prints rel-L2 per stage for the tiny fixtures
and (with --real) the real OWL-ViT / SAM widths.  usage: python tools/mask_head_probe.py [--real]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import vsm_oracle  # noqa: E402  (diagnostic tool, not product)
from vstar_amd.config import VSMConfig  # noqa: E402
from vstar_amd.engine import VstarEngine, loc_positions  # noqa: E402
from vstar_amd.weights import random_state_dict  # noqa: E402


def rel(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def run(cfg, wseed, B, L, seed):
    loc_id = cfg.llm_vocab - 1
    sd = random_state_dict(cfg, seed=wseed, dtype=torch.bfloat16)
    eng = VstarEngine(cfg, 0)
    eng.load_state_dict(sd)
    g = torch.Generator().manual_seed(seed)
    I = cfg.clip_image_size
    clip = torch.randn(B, 3, I, I, generator=g).bfloat16()
    owl = torch.randn(B, 3, 768, 768, generator=g).bfloat16()
    ids = torch.randint(3, loc_id - 3, (B, L), generator=g)
    ids[:, 0] = 1
    ids[:, 5] = -200
    ids[:, L - 3] = loc_id
    loc = loc_positions(ids.numpy(), loc_id, cfg.n_img_tokens)
    out = eng.score_batch(clip, owl, ids.numpy(), loc)
    sd32 = {k: v.float() for k, v in sd.items()}
    ref = vsm_oracle.vsm_forward(sd32, cfg, clip.float(), owl.float(), ids, loc_id)
    refb = vsm_oracle.vsm_forward(sd, cfg, clip, owl, ids, loc_id)          # the same algorithm in bf16 on torch-CPU
    NP = 2304
    sizes = {"sam_src": NP * 256, "sam_tokens": 6 * 256, "sam_keys": NP * 256, "sam_c1": 96 * 96 * 64, "sam_c1n": 96 * 96 * 64,
             "sam_c2": 192 * 192 * 32, "sam_hyper": 32}
    print(f"{'stage':<14s} engine-vs-fp32   bf16-torch-vs-fp32")
    for name in ("owl_feats", "embed_seg"):
        n = B * (NP * cfg.owl_hidden if name == "owl_feats" else 256)
        print(f"{name:<14s} {rel(eng.debug_read(name, n), ref[name].numpy()):.2e}         {rel(refb[name].float().numpy(), ref[name].numpy()):.2e}")
    for name, n in sizes.items():
        got = eng.debug_read(name, B * n)
        print(f"{name:<14s} {rel(got, ref['sam_taps'][name].numpy()):.2e}         "
              f"{rel(refb['sam_taps'][name].float().numpy(), ref['sam_taps'][name].numpy()):.2e}")
    for b in range(B):
        print(f"mask crop {b}    {rel(out['low_res_masks'][b], ref['low_res_masks'][b].numpy()):.2e}         "
              f"{rel(refb['low_res_masks'][b].float().numpy(), ref['low_res_masks'][b].numpy()):.2e}")
    # mask recomputed in fp64 from the ENGINE's own c2 / hyper taps: isolates the final 32-term product
    c2 = eng.debug_read("sam_c2", B * sizes["sam_c2"]).reshape(B, -1, 32).astype(np.float64)
    hy = eng.debug_read("sam_hyper", B * 32).reshape(B, 32).astype(np.float64)
    m = np.einsum("bpc,bc->bp", c2, hy)
    print("final product alone (engine mask vs fp64 product of engine taps):", ["%.2e" % rel(out["low_res_masks"][b], m[b]) for b in range(B)])
    # which operand carries the mask error?  fp64 products mixing engine and oracle operands
    rc2 = ref["sam_taps"]["sam_c2"].numpy().astype(np.float64)
    rhy = ref["sam_taps"]["sam_hyper"].numpy().astype(np.float64)
    bc2 = refb["sam_taps"]["sam_c2"].float().numpy().astype(np.float64)
    bhy = refb["sam_taps"]["sam_hyper"].float().numpy().astype(np.float64)
    full = np.einsum("bpc,bc->bp", rc2, rhy)
    for tag, (C, Hh) in {"engine c2 x oracle hyper": (c2, rhy), "oracle c2 x engine hyper": (rc2, hy),
                         "bf16-torch c2 x oracle hyper": (bc2, rhy), "oracle c2 x bf16-torch hyper": (rc2, bhy)}.items():
        print(f"  {tag:<30s}", ["%.2e" % rel(np.einsum("pc,c->p", C[b], Hh[b]), full[b]) for b in range(B)])
    for tag, C in (("engine", c2), ("bf16-torch", bc2)):
        d = C - rc2
        print(f"  c2 error structure [{tag}]: rel {rel(C, rc2):.2e}, mean(d)/rms(d) per channel max "
              f"{np.abs(d.mean(axis=1) / np.sqrt((d ** 2).mean(axis=1))).max():.2f}, cancellation |sum|/sum|.| of the product "
              f"{np.abs(full).mean() / np.einsum('bpc,bc->bp', np.abs(rc2), np.abs(rhy)).mean():.3f}")
    # hypernetwork MLP re-evaluated in fp64 from the ENGINE's own token tap: separates the MLP's arithmetic from its input error
    md = "model.mask_decoder.output_hypernetworks_mlps.0.layers."
    tok = eng.debug_read("sam_tokens", B * 6 * 256).reshape(B, 6, 256)[:, 1].astype(np.float64)
    rtok = ref["sam_taps"]["sam_tokens"].numpy()[:, 1].astype(np.float64)
    btok = refb["sam_taps"]["sam_tokens"].float().numpy()[:, 1].astype(np.float64)

    def mlp(t):
        for j in range(3):
            t = t @ sd32[md + f"{j}.weight"].double().numpy().T + sd32[md + f"{j}.bias"].double().numpy()
            if j < 2:
                t = np.maximum(t, 0)
        return t
    mu = rc2.mean(axis=1)                                              # [B,32] common-mode direction of the upscaled features
    print("  |mu|/rms|c2(p)-mu| =", ["%.2f" % (np.linalg.norm(mu[b]) / np.sqrt(((rc2[b] - mu[b]) ** 2).sum(1).mean())) for b in range(B)])
    for tag, hh, tt in (("engine", hy, tok), ("bf16-torch", bhy, btok)):
        ex = mlp(tt)
        print(f"  hyper [{tag}]: vs fp64 MLP of its OWN tokens {rel(hh, ex):.2e}; fp64 MLP of its tokens vs oracle {rel(ex, rhy):.2e}; "
              f"token err {rel(tt, rtok):.2e}; (dh.mu)/(h.mu) = {[('%.3f' % (((hh - rhy)[b] @ mu[b]) / (rhy[b] @ mu[b]))) for b in range(B)]}")
    eng.close()


if __name__ == "__main__":
    if "--seeds" in sys.argv:
        for ws in range(20, 26):
            print("== weight seed", ws)
            run(VSMConfig.seal_7b(224, clip_layers=1, llm_layers=1, owl_layers=2, llm_vocab=1024, max_batch=2, max_text_len=24), ws, 2, 24, ws)
    elif "--real" in sys.argv:
        run(VSMConfig.seal_7b(224, clip_layers=2, llm_layers=1, owl_layers=12, llm_vocab=1024, max_batch=2, max_text_len=24), 11, 2, 24, 4)
    else:
        run(VSMConfig.tiny(), 0, 2, 24, 1)
        run(VSMConfig.tiny(clip_image_size=336), 3, 2, 20, 21)
