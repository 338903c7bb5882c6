"""Micro-benchmark of the VQA-LLM engine at the 7B geometry (seeded random fp16 weights): image encoding, prefill,
KV-cached decode steps at several batch sizes, forked option scoring.  Prints one JSON object.

  python tools/vqa_bench.py [--out profiles/r01_vqa_bench.json] [--layers 32]
Decode steps are bound by the weight sweep (13.5 GB fp16 per step at 7B): `weights_GBps` = bytes of all LLaMA + lm_head
weights / device time of one step (HIP events inside the engine), against the ~8 TB/s HBM3E peak.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vstar_amd.config import VQAConfig  # noqa: E402
from vstar_amd.vqa_engine import Seq, VqaEngine  # noqa: E402
from vstar_amd.weights import random_state_dict  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--batches", default="1,4,16,32")
    ap.add_argument("--steps", type=int, default=16)
    a = ap.parse_args()
    cfg = VQAConfig.seal_7b(llm_layers=a.layers, max_slots=40, max_ctx=1024, max_rows=16384, max_images=8)
    t0 = time.time()
    eng = VqaEngine(cfg, 0)
    eng.load_state_dict(random_state_dict(cfg, 0, torch.float16, share_layers=True))
    load_s = time.time() - t0
    H, M, V, L = cfg.llm_hidden, cfg.llm_mlp, cfg.llm_vocab, cfg.llm_layers
    wbytes = 2.0 * (L * (4 * H * H + 3 * H * M) + V * H)
    out = {"config": {"layers": L, "hidden": H, "vocab": V, "weights_GB": round(wbytes / 1e9, 2)}, "weights_load_s": round(load_s, 1)}
    g = torch.Generator().manual_seed(0)
    pix = torch.randn(3, 3, 224, 224, generator=g)
    eng.encode_images(pix, 0)
    t0 = time.time()
    for _ in range(3):
        eng.encode_images(pix, 0)
    out["encode_3_images_ms"] = round((time.time() - t0) / 3 * 1e3, 2)
    # prompt: 40 text ids + 1 long image (256 rows) + 2 long objects (512 rows) = 808 rows (the <=2-objects case), or short
    text = torch.randint(3, 30000, (40,), generator=g).tolist()
    long_rows = text[:10] + eng.feature_rows(0, True) + text[10:25] + eng.feature_rows(1, True) + eng.feature_rows(2, True) + text[25:]
    short_rows = text[:10] + eng.feature_rows(0, False) + text[10:25] + eng.feature_rows(1, True) + eng.feature_rows(2, True) + text[25:]
    plain_rows = text[:10] + eng.feature_rows(0, True) + text[10:]
    res = {}
    for name, rows in (("plain_296", plain_rows), ("objects_584", short_rows)):
        for B in (1, 8):
            seqs = [Seq(rows, kv_slot=i) for i in range(B)]
            eng.forward(seqs, [(i, -1) for i in range(B)], logits=False)
            ms = []
            for _ in range(3):
                eng.forward(seqs, [(i, -1) for i in range(B)], logits=False)
                ms.append(eng.last_forward_ms())
            S = len(rows)
            flops = 2.0 * B * S * (L * (4 * H * H + 3 * H * M)) + 2.0 * B * L * S * S * H
            res[f"prefill_{name}_B{B}"] = {"ms": round(min(ms), 3), "TFLOPs": round(flops / min(ms) / 1e9, 1),
                                           "rows": B * S}
    out["prefill"] = res
    # decode: B sequences with ~300 cached positions each
    dec = {}
    for B in [int(x) for x in a.batches.split(",")]:
        seqs = [Seq(plain_rows, kv_slot=i) for i in range(B)]
        _, nxt = eng.forward(seqs, [(i, -1) for i in range(B)], logits=False)
        pos = len(plain_rows)
        dev, t0 = [], time.time()
        for s in range(a.steps):
            _, nxt = eng.forward([Seq([int(nxt[i])], kv_slot=i, past_len=pos) for i in range(B)], [(i, 0) for i in range(B)],
                                 logits=False)
            dev.append(eng.last_forward_ms())
            pos += 1
        wall = (time.time() - t0) / a.steps * 1e3
        d = float(np.median(dev))
        dec[f"B{B}"] = {"device_ms_per_step": round(d, 3), "wall_ms_per_step": round(wall, 3),
                        "tokens_per_s": round(B / wall * 1e3, 1), "weights_GBps": round(wbytes / d / 1e6, 0)}
    out["decode"] = dec
    # option scoring: 4 options x 12 tokens forked from one question prefix
    eng.forward([Seq(short_rows, kv_slot=0)], [(0, -1)])
    P = len(short_rows)
    opts = [torch.randint(3, 30000, (12,), generator=g).tolist() for _ in range(4)]
    ms = []
    for _ in range(4):
        eng.forward([Seq(o, kv_slot=1 + j, past_len=P, prefix_slot=0) for j, o in enumerate(opts)],
                    [(j, t) for j in range(4) for t in range(11)])
        ms.append(eng.last_forward_ms())
    out["option_scoring_4x12_ms"] = round(min(ms), 3)
    print(json.dumps(out))
    if a.out:
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
