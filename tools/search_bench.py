#!/usr/bin/env python
"""End-to-end search benchmark (BASELINE config 2 shape): one synthetic 4K image, exhaustive depth-3 search tree
(1 + 4 + 16 = 21 crops per target), 7B VSM with seeded random weights, crops scored in engine batches of 32.
Reports wall-clock crops/s of the WHOLE loop (host PIL preprocessing, H2D, engine, record D2H, heatmap upsample, decisions)
and where the time goes.  Not the headline metric (bench.py keeps inputs HBM-resident); it shows what the next scope rows
(GPU-side preprocessing, on-device reductions) are worth."""
import argparse
import json
import os
import sys
import time
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vstar_amd.synthetic import synthetic_image  # noqa: E402
from vstar_amd.config import VSMConfig  # noqa: E402
from vstar_amd.preprocess import SyntheticTokenizer  # noqa: E402
from vstar_amd.search import smallest_size_for, visual_search, visual_search_many  # noqa: E402
from vstar_amd.vsm import VSM  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--targets", type=int, default=4)
ap.add_argument("--image-size", type=int, default=336)
ap.add_argument("--tiny", action="store_true")
ap.add_argument("--device-reductions", action="store_true")
ap.add_argument("--host-preprocess", action="store_true")
ap.add_argument("--many", action="store_true", help="visual_search_many: first step of all targets batched together")
args = ap.parse_args()
cfg = (VSMConfig.tiny if args.tiny else VSMConfig.seal_7b)(clip_image_size=args.image_size, max_batch=32, max_text_len=128) \
    if args.tiny else VSMConfig.seal_7b(args.image_size, max_batch=32, max_text_len=128)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    vsm = VSM(None, cfg=cfg, tokenizer=SyntheticTokenizer(cfg.llm_vocab), synthetic_seed=0, strict_template=False)
    img = synthetic_image(3840, 2160, 0)
    smallest = smallest_size_for(3840, 2160)
    kw = dict(confidence_high=2.0, confidence_low=0.0, target_cue_threshold=-1.0, target_cue_threshold_minimum=-1.0,
              device_reductions=args.device_reductions, gpu_preprocess=not args.host_preprocess)
    visual_search(vsm, img, "warmup", None, smallest, **kw)
    for k in vsm.timers:
        vsm.timers[k] = 0
    t0 = time.perf_counter()
    tot = {"crops_scored": 0, "engine_batches": 0, "path_visited": 0}
    if args.many:
        visual_search_many(vsm, img, [f"object {i}" for i in range(args.targets)], None, smallest, **kw)
        tot["crops_scored"] = vsm.timers["crops"]
    else:
        for i in range(args.targets):
            st = {}
            visual_search(vsm, img, f"object {i}", None, smallest, stats=st, **kw)
            for k in tot:
                tot[k] += st[k]
    dt = time.perf_counter() - t0
print(json.dumps({"mode": {"gpu_preprocess": not args.host_preprocess, "device_reductions": args.device_reductions},
                  "search_crops_per_s": round(tot["crops_scored"] / dt, 2), "wall_s": round(dt, 3), **tot,
                  "timers": {k: round(v, 3) if isinstance(v, float) else v for k, v in vsm.timers.items()},
                  "decision_and_other_s": round(dt - sum(v for k, v in vsm.timers.items() if k.endswith("_s")), 3)}))
