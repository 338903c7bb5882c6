"""Per-crop spread of the OWL-ViT tower's bf16 error on the trained-like bench batch (round 4).

The full-depth trained-like golden (tests/golden/full7b_tl_336.npz) has ONE crop (crop 0) whose OWL-dependent taps sit at 1.4 - 1.7 x
the reference-bf16 noise while the other crops — and every LLaMA tap of that crop — sit at 1.0 x.  OwlViT multiplies every patch
token by the class token (owlvit.py:128-138), so the error of ONE row (the CLS token) scales a whole crop's features: the per-crop
error is a heavy-tailed statistic.  This tool measures the whole distribution: all 32 crops of the batch through the engine and
through the fp32 / bf16 oracle (torch on the GPU), per-crop rel-L2 of `owl_feats` for both.  GPU only; test infrastructure."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import vsm_oracle  # noqa: E402
from vstar_amd.config import VSMConfig  # noqa: E402
from vstar_amd.engine import VstarEngine  # noqa: E402
from vstar_amd.preprocess import SyntheticTokenizer  # noqa: E402
from vstar_amd.synthetic import bench_inputs  # noqa: E402
from vstar_amd.weights import template_chain, trained_like_state_dict  # noqa: E402


def rel(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def main():
    dev = torch.device("cuda", 0)
    B, T = 32, 64
    cfg = VSMConfig.seal_7b(336, max_batch=B, max_text_len=T + 1)
    sd16 = trained_like_state_dict(cfg, seed=0, dtype=torch.bfloat16, share_layers=True, chain=template_chain(SyntheticTokenizer(cfg.llm_vocab)))
    clip, owl, ids, loc, verify = bench_inputs(cfg, B, T)
    eng = VstarEngine(cfg, 0)
    eng.load_state_dict(sd16)
    eng.score_batch(clip.to(dev), owl.to(dev), ids, loc, verify_pos=verify)
    fe = eng.debug_read("owl_feats", B * 2304 * 768).reshape(B, 2304, 768)
    eng.close()
    out = {"engine": [], "bf16": [], "cls_bf16": []}
    sds = {dt: {k: v.to(dev, dt) for k, v in sd16.items() if k.startswith("model.owlvit")} for dt in (torch.float32, torch.bfloat16)}
    for b in range(B):
        with torch.no_grad():
            f32 = vsm_oracle.owl_visual_embs(sds[torch.float32], owl[b:b + 1].to(dev, torch.float32), cfg.owl_heads, cfg.owl_layers)
            f16 = vsm_oracle.owl_visual_embs(sds[torch.bfloat16], owl[b:b + 1].to(dev, torch.bfloat16), cfg.owl_heads, cfg.owl_layers)
        r32 = f32.float().reshape(2304, 768).cpu().numpy()
        out["engine"].append(rel(fe[b], r32))
        out["bf16"].append(rel(f16.float().reshape(2304, 768).cpu().numpy(), r32))
        print(f"crop {b:2d}  engine {out['engine'][-1]:.3e}  torch-bf16 {out['bf16'][-1]:.3e}  ratio {out['engine'][-1] / out['bf16'][-1]:.2f}", flush=True)
    e, n = np.asarray(out["engine"]), np.asarray(out["bf16"])
    summ = {"crops": B, "engine_median": float(np.median(e)), "bf16_median": float(np.median(n)), "engine_max": float(e.max()),
            "bf16_max": float(n.max()), "engine_rms": float(np.sqrt((e ** 2).mean())), "bf16_rms": float(np.sqrt((n ** 2).mean())),
            "engine_max_over_median": float(e.max() / np.median(e)), "bf16_max_over_median": float(n.max() / np.median(n)),
            "per_crop_ratio_min_max": [float((e / n).min()), float((e / n).max())], "per_crop": out}
    print(json.dumps({k: v for k, v in summ.items() if k != "per_crop"}, indent=1))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(summ, open(os.path.join(ROOT, "gpurun_out", "owl_crop_spread.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
