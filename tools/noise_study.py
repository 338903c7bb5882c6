#!/usr/bin/env python
"""Where does the engine's numerical noise sit next to the reference's own bf16 run?  

Scores the 32-crop bench batch (trained-like weights, CLIP@336, S = 640) and compares EVERY crop with
tests/golden/full7b_tl_336_x32.npz — the reference's model_forward(inference=True) in fp32 and in bf16 for all 32 crops
(oracle/gen_fulldepth_golden.py --weights trained_like --crops all --out full7b_tl_336_x32.npz --mask-f16; round 4 had 8 crops, and a
ratio of two rms values over 8 heavy-tailed per-crop errors moves by +-20 % from sampling alone).  Per tap it prints, pooled (rms)
over the 32 crops: engine vs fp32, reference-bf16 vs fp32, their ratio, and the DIRECT distance engine <-> reference-bf16 (two
independent roundings of one fp32 value are sqrt(2) x one rounding apart; a systematic difference shows as more).

One configuration per process (the engine reads its switches at creation); `--sweep` runs the bisect matrix as subprocesses:
  default | VSTAR_FOLD_ZERO_SUM=0 | VSTAR_FOLD_VIT_NORMS=0 | VSTAR_FOLD_NORMS=0 | exact-sigmoid build (VSTAR_LIB=...lib_exactsig.so)
usage: python tools/noise_study.py --sweep --out gpurun_out/r05/noise_study.json
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
GEOM = int(os.environ.get("NOISE_STUDY_GEOMETRY", "336"))          # 336 (bench geometry) or 224 (the reference's own)
GOLD = os.path.join(ROOT, "tests", "golden", f"full7b_tl_{GEOM}_x32.npz")


def one():
    import numpy as np
    import torch
    from _parity import assert_mask_within_bf16_noise, rel_l2
    from vstar_amd.config import VSMConfig
    from vstar_amd.engine import VstarEngine
    from vstar_amd.preprocess import SyntheticTokenizer
    from vstar_amd.synthetic import bench_inputs
    from vstar_amd.weights import template_chain, trained_like_state_dict
    z = np.load(GOLD)
    B, T = int(z["batch"]), int(z["text_tokens"])
    cfg = VSMConfig.seal_7b(GEOM, max_batch=B, max_text_len=T + 1)
    eng = VstarEngine(cfg, 0)
    eng.load_state_dict(trained_like_state_dict(cfg, seed=int(z["weight_seed"]), dtype=torch.bfloat16, share_layers=True,
                                                chain=template_chain(SyntheticTokenizer(cfg.llm_vocab))))
    clip, owl, ids, loc, verify = bench_inputs(cfg, B, T)
    out = eng.score_batch(clip.cuda(), owl.cuda(), ids, loc, verify_pos=verify)
    H = cfg.llm_hidden
    taps = {"llm_hidden_loc": eng.debug_read("llm_hidden_loc", B * H).reshape(B, H),
            "embed_det": eng.debug_read("embed_det", B * 512).reshape(B, 512),
            "embed_seg": eng.debug_read("embed_seg", B * 256).reshape(B, 256),
            "sam_hyper": eng.debug_read("sam_hyper", B * 32).reshape(B, 32),
            "pred_logits": out["pred_logits"][:, :, 0], "pred_boxes": out["pred_boxes"]}
    c2 = eng.debug_read("sam_c2", B * 192 * 192 * 32).reshape(B, -1, 32).astype(np.float64).mean(axis=1)
    taps["sam_upscaled_mean"] = c2
    rms = lambda xs: float(np.sqrt(np.mean(np.square(xs))))  # noqa: E731
    rep = {}
    crops = [int(c) for c in z["crops"]]
    for k, got in taps.items():
        e = [rel_l2(got[ci], z[k][j]) for j, ci in enumerate(crops)]
        n = [rel_l2(z["bf16_" + k][j], z[k][j]) for j in range(len(crops))]
        d = [rel_l2(got[ci], z["bf16_" + k][j]) for j, ci in enumerate(crops)]
        rep[k] = {"engine": rms(e), "ref_bf16": rms(n), "ratio": rms(e) / rms(n), "worst_crop_ratio": float(np.max(np.asarray(e) / np.asarray(n))),
                  "engine_vs_ref_bf16": rms(d), "direct_over_sqrt2_noise": rms(d) / (np.sqrt(2.0) * rms(n))}
    pat = lambda m: m - m.mean(axis=(1, 2), keepdims=True)  # noqa: E731
    m32, m16 = z["low_res_masks"].astype(np.float64), z["bf16_low_res_masks"].astype(np.float64)
    mg = out["low_res_masks"][crops, 0].astype(np.float64)
    e = [rel_l2(pat(mg[j:j + 1]), pat(m32[j:j + 1])) for j in range(len(crops))]
    n = [rel_l2(pat(m16[j:j + 1]), pat(m32[j:j + 1])) for j in range(len(crops))]
    rep["mask_pattern"] = {"engine": rms(e), "ref_bf16": rms(n), "ratio": rms(e) / rms(n), "worst_crop_ratio": float(np.max(np.asarray(e) / np.asarray(n)))}
    rep["mask_uncentred"] = {"engine": rel_l2(mg, m32), "ref_bf16": rel_l2(m16, m32)}
    rep["mask_uncentred"]["ratio"] = rep["mask_uncentred"]["engine"] / rep["mask_uncentred"]["ref_bf16"]
    zs_e, zs_n = [], []
    for j, ci in enumerate(crops):
        r = {}
        try:
            assert_mask_within_bf16_noise(mg[j], m32[j], m16[j], z["sam_hyper"][j], z["bf16_sam_hyper"][j], z["sam_upscaled_mean"][j],
                                          z["bf16_sam_upscaled_mean"][j], factor=1e9, report=r)
        except AssertionError:
            pass
        zs_e.append(r["mask_offset_sigma[0]"][0]); zs_n.append(r["mask_offset_sigma[0]"][1])
    rep["mask_offset_sigma"] = {"engine_rms": rms(zs_e), "engine_max": float(np.max(zs_e)), "ref_bf16_rms": rms(zs_n), "ref_bf16_max": float(np.max(zs_n))}
    # signed offsets: a systematic bias shows as a non-zero mean
    off_e = [float(mg[j].mean() - m32[j].mean()) for j in range(len(crops))]
    off_n = [float(m16[j].mean() - m32[j].mean()) for j in range(len(crops))]
    rep["mask_offset_signed"] = {"engine_mean": float(np.mean(off_e)), "engine_std": float(np.std(off_e)), "ref_bf16_mean": float(np.mean(off_n)),
                                 "ref_bf16_std": float(np.std(off_n))}
    eng.close()
    print("NOISE_STUDY " + json.dumps(rep))


def sweep(out_path):
    ab = os.path.join(ROOT, "vstar_amd", "csrc", "build", "ab")
    configs = [("default", {}), ("fold_zero_sum_off", {"VSTAR_FOLD_ZERO_SUM": "0"}), ("vit_fold_off", {"VSTAR_FOLD_VIT_NORMS": "0"}),
               ("all_folds_off", {"VSTAR_FOLD_NORMS": "0"})]
    for name in sorted(os.listdir(ab)) if os.path.isdir(ab) else []:
        if name.startswith("lib_ns_") and name.endswith(".so"):
            configs.append((name[7:-3], {"VSTAR_LIB": os.path.join(ab, name)}))
    res = {}
    for name, env in configs:
        p = subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
        line = [l for l in p.stdout.splitlines() if l.startswith("NOISE_STUDY ")]
        res[name] = json.loads(line[-1][12:]) if line else {"error": p.stderr[-800:]}
        if line:
            r = res[name]
            print(f"{name:22s} " + "  ".join(f"{k} x{v['ratio']:.2f}" for k, v in r.items() if "ratio" in v) +
                  f"  offset rms {r['mask_offset_sigma']['engine_rms']:.2f} (ref {r['mask_offset_sigma']['ref_bf16_rms']:.2f})", flush=True)
        else:
            print(name, "FAILED", p.stderr[-400:], flush=True)
    if out_path:
        json.dump(res, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--sweep", action="store_true")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    sweep(a.out) if a.sweep else one()
