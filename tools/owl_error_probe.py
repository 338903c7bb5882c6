"""Where does the OWL-ViT side lose accuracy on trained-like weights?  (round 4: pred_boxes landed at 1.51 x the reference-bf16
noise on the trained-like full-depth golden, every other tap at 1.0 x.)

Runs the real OWL-ViT-B/16@768 geometry (tiny LLaMA / CLIP: they do not feed the boxes' tower) at several depths and weight-feature
sets and prints rel-L2 against the fp32 oracle (torch on the GPU) for the engine and for the same oracle in bf16 — the yardstick
of tests/_parity.py — on `owl_feats` and `pred_boxes`.  GPU only; test infrastructure (imports oracle/)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import vsm_oracle  # noqa: E402
from vstar_amd.config import VSMConfig  # noqa: E402
from vstar_amd.engine import VstarEngine  # noqa: E402
from vstar_amd.weights import random_state_dict, trained_like_state_dict  # noqa: E402


def rel(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def main():
    dev = torch.device("cuda", 0)
    B = 4
    rows = []
    for feats in (("outliers", "attn", "norms"), ("outliers",), ("attn",), ("norms",), ()):
        for L in (1, 4, 12):
            cfg = VSMConfig.tiny(owl_hidden=768, owl_heads=12, owl_mlp=3072, owl_layers=L, max_batch=B, max_text_len=32)
            if feats:
                sd16 = trained_like_state_dict(cfg, seed=0, dtype=torch.bfloat16, share_layers=True, features=feats)
            else:
                sd16 = random_state_dict(cfg, seed=0, dtype=torch.bfloat16, share_layers=True)
            g = torch.Generator().manual_seed(5)
            clip = torch.randn(B, 3, 224, 224, generator=g).bfloat16()
            owl = torch.randn(B, 3, 768, 768, generator=g).bfloat16()
            L_txt = 20
            ids = np.random.default_rng(0).integers(3, cfg.llm_vocab - 5, size=(B, L_txt), dtype=np.int32)
            ids[:, 0], ids[:, 2], ids[:, L_txt - 3] = 1, -200, cfg.llm_vocab - 1
            loc = np.full((B,), (L_txt - 3) - 1 + (cfg.n_img_tokens - 1), np.int32)
            eng = VstarEngine(cfg, 0)
            eng.load_state_dict(sd16)
            out = eng.score_batch(clip.to(dev), owl.to(dev), ids, loc)
            feats_e = eng.debug_read("owl_feats", B * 2304 * 768).reshape(B, 2304, 768)
            det_e = eng.debug_read("embed_det", B * 512).reshape(B, 512)
            eng.close()
            res = {}
            for name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
                sd = {k: v.to(dev, dt) for k, v in sd16.items() if k.startswith("model.owlvit")}
                with torch.no_grad():
                    fm = vsm_oracle.owl_visual_embs(sd, owl.to(dev, dt), cfg.owl_heads, cfg.owl_layers)
                    # the engine's own query embedding for both: isolates the tower + heads from the LLaMA path
                    q = torch.from_numpy(det_e).to(dev, dt).unsqueeze(1)
                    lg, bx = vsm_oracle.owl_heads(sd, fm, q)
                res[name] = (fm.float().reshape(B, 2304, 768).cpu().numpy(), bx.float().cpu().numpy(), lg.float().cpu().numpy())
            row = {"features": "+".join(feats) or "random", "layers": L,
                   "owl_feats": (rel(feats_e, res["f32"][0]), rel(res["bf16"][0], res["f32"][0])),
                   "pred_boxes": (rel(out["pred_boxes"], res["f32"][1]), rel(res["bf16"][1], res["f32"][1])),
                   "pred_logits": (rel(out["pred_logits"], res["f32"][2]), rel(res["bf16"][2], res["f32"][2]))}
            rows.append(row)
            print(f"{row['features']:24s} L={L:2d}  " + "  ".join(
                f"{k} {row[k][0]:.2e}/{row[k][1]:.2e} (x{row[k][0] / max(row[k][1], 1e-30):.2f})" for k in ("owl_feats", "pred_boxes", "pred_logits")),
                flush=True)
    import json
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "owl_error_probe.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
