#!/usr/bin/env python
"""Search-only evaluation entry point (same flags and printed metrics as the reference's visual_search.py:28-52,520-566),
driven by the HIP engine: `VSM` from vstar_amd.vsm, batched `visual_search` from vstar_amd.search.

  python visual_search.py --version /path/to/seal_vsm_7b --vision-tower /path/to/clip-vit-large-patch14 \
         --benchmark-folder vstar_bench
Extra (additive) flags: --device, --batch, --synthetic-seed (random weights when no checkpoint is staged), --shard, --window.

The reference searches ONE (image, target) at a time (visual_search.py:536-560).  Here `--window K` samples are searched in lock
step (vstar_amd.search.visual_search_stream): every engine batch holds the crops K concurrent searches need next — crops of different
images side by side through the engine's image slots — with per-sample results identical to the one-at-a-time loop.  --window 1 is the
reference's schedule (plus cost-aware speculation of a lone search's likely next crops).

Multi-GPU (BASELINE configs 3/4): launch with torchrun, one process per GPU —
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 visual_search.py ...
--shard crops   (default; the north-star layout): every rank walks the same samples; each engine step's crop batch is dealt
                round-robin over the ranks and the fixed-size records are all-gathered over RCCL, so every rank takes the same
                next-step decision.
--shard samples : (image, target) pairs are dealt round-robin over the ranks, each search runs on one GPU, the per-sample
                metrics are gathered at the end (no collective on the data path).
Rank 0 prints the metrics, which are identical to a single-process run in both modes.
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import numpy as np
from PIL import Image

from vstar_amd.config import VSMConfig
from vstar_amd.dist import finalize, gather_objects, init_from_env
from vstar_amd.search import iou, smallest_size_for, visual_search_stream

SPLITS = ("direct_attributes", "relative_position")


def parse_args(argv):
    p = argparse.ArgumentParser(description="Visual Search Evaluation")
    p.add_argument("--version", default="craigwu/seal_vsm_7b")
    p.add_argument("--benchmark-folder", default="vstar_bench", type=str)
    p.add_argument("--visualization", action="store_true", default=False)
    p.add_argument("--output_path", default="", type=str)
    p.add_argument("--confidence_low", default=0.3, type=float)
    p.add_argument("--confidence_high", default=0.5, type=float)
    p.add_argument("--target_cue_threshold", default=6.0, type=float)
    p.add_argument("--target_cue_threshold_decay", default=0.7, type=float)
    p.add_argument("--target_cue_threshold_minimum", default=3.0, type=float)
    p.add_argument("--minimum_size_scale", default=4.0, type=float)
    p.add_argument("--minimum_size", default=224, type=int)
    p.add_argument("--model_max_length", default=512, type=int)
    p.add_argument("--vision-tower", default="openai/clip-vit-large-patch14", type=str)
    p.add_argument("--use_mm_start_end", action="store_true", default=True)
    p.add_argument("--conv_type", default="llava_v1", type=str, choices=["llava_v1", "llava_llama_2"])
    # additive
    p.add_argument("--device", default=0, type=int)
    p.add_argument("--batch", default=32, type=int, help="crops per engine pass")
    p.add_argument("--synthetic-seed", default=None, type=int)
    p.add_argument("--shard", default="crops", choices=["crops", "samples", "auto"], help="what is dealt over the ranks under torchrun; "
                   "auto = vstar_amd.dist.choose_shard (measured step-time table: samples when every rank can be kept busy with whole "
                   "searches, crops when there are fewer searches than ranks x window)")
    p.add_argument("--window", default=0, type=int, help="concurrent (image, target) searches per process; 0 = one engine batch "
                   "(x world size when crops are sharded); 1 = the reference's one-sample-at-a-time schedule")
    p.add_argument("--engine-comm", nargs="?", const="on", default="auto", choices=["auto", "on", "off"],
                   help="world > 1 on GPUs, crop sharding: gather the per-step records with the C-ABI's own RCCL collective "
                   "(vstar_allgather_results) instead of torch.distributed.  auto (default, round 6) = use it when the communicator comes "
                   "up and its self-check against torch.distributed passes on every rank, else fall back; on = the same, and the JSON "
                   "summary records the outcome; off = torch.distributed")
    p.add_argument("--vsm-factory", default=None, help="module:factory(args, device) returning an object with the VSM interface "
                   "(tests substitute a CPU stand-in for the engine)")
    return p.parse_args(argv)


def iter_samples(folder):
    for split in SPLITS:
        d = os.path.join(folder, split)
        for name in sorted(os.listdir(d)):
            if name.endswith(".json"):
                continue
            ann = json.load(open(os.path.join(d, os.path.splitext(name)[0] + ".json")))
            for k, (gt_bbox, target) in enumerate(zip(ann["bbox"], ann["target_object"])):
                yield split, os.path.join(d, name), gt_bbox, target, k


def make_vsm(args, device):
    if args.vsm_factory:
        import importlib
        mod, fn = args.vsm_factory.split(":")
        return getattr(importlib.import_module(mod), fn)(args, device)
    from vstar_amd.vsm import VSM
    return VSM(args, cfg=VSMConfig.seal_7b(224, max_batch=args.batch), device=device, synthetic_seed=args.synthetic_seed)


def main(argv):
    args = parse_args(argv)
    if args.visualization and not args.output_path:
        raise SystemExit("--visualization needs --output_path (the directories are written beneath it, visual_search.py:531-548)")
    world, rank, local_rank = init_from_env()
    finished = False
    try:
        vsm = make_vsm(args, local_rank if world > 1 else args.device)
        if args.shard == "auto":                         # the same decision on every rank: it depends on the sample list, world and window only
            from vstar_amd.dist import choose_shard
            n_all = sum(1 for _ in iter_samples(args.benchmark_folder))
            args.shard = choose_shard(n_all, world, args.window or args.batch)
            if rank == 0:
                print(f"--shard auto: {n_all} searches on {world} rank(s), window {args.window or args.batch} -> {args.shard}")
        if args.shard == "samples":
            vsm.shard_crops = False                      # each search stays on its own GPU
        elif args.shard == "crops" and args.engine_comm != "off" and world > 1:
            from vstar_amd.dist import maybe_engine_comm
            maybe_engine_comm(vsm)                       # (no-op on gloo / CPU; falls back to torch.distributed on any failure)
        class _Loader:                                   # lazy image load when the sample enters the window; one slot per file
            def __init__(self, path):
                self.key = path

            def __call__(self):
                return Image.open(self.key).convert("RGB")

        mine, loaders = [], {}
        vis_dirs = {}
        for i, (split, path, gt_bbox, target, k) in enumerate(iter_samples(args.benchmark_folder)):
            if args.shard == "samples" and i % world != rank:
                continue
            if args.visualization:                       # visual_search.py:531-548: <output_path>/<split>/<image stem>_<k>/
                vis_dirs[i] = os.path.join(args.output_path, split, "{}_{}".format(os.path.basename(path).split(".")[0], k))
            ld = loaders.setdefault(path, _Loader(path))
            mine.append((i, gt_bbox, (ld, target, gt_bbox,
                                      lambda im: smallest_size_for(im.width, im.height, args.minimum_size_scale, args.minimum_size))))
        stats = {}
        skw = dict(confidence_high=args.confidence_high, confidence_low=args.confidence_low,
                   target_cue_threshold=args.target_cue_threshold, target_cue_threshold_decay=args.target_cue_threshold_decay,
                   target_cue_threshold_minimum=args.target_cue_threshold_minimum)
        if args.visualization:
            # the reference's one-sample-at-a-time loop (visual_search.py:536-550): every search renders its own directory; the
            # full-resolution heat maps are materialised on the host for it (no device reductions)
            from vstar_amd.search import visual_search
            outs = []
            for i, _, (ld, target, gt_bbox, small_of) in mine:
                im = ld()
                # every rank takes the SAME decision path (host float32 heat maps, no device reductions); with crop sharding all
                # ranks walk every search, so only rank 0 renders (rank 0 alone used to leave the device-reduction path)
                render = rank == 0 or args.shard == "samples"
                outs.append(visual_search(vsm, im, target, gt_bbox, small_of(im), visualize=render,
                                          save_path=vis_dirs[i], device_reductions=False, **skw))
        else:
            outs = visual_search_stream(vsm, [m[2] for m in mine], window=args.window or None, stats=stats, **skw)
        results = []                                     # (sample index, hit, path length)
        for (i, gt_bbox, _), (step, n_steps, ok, _) in zip(mine, outs):
            if not ok:
                results.append((i, 0.0, 0))
                continue
            box = step["detection_result"]
            box[0] += step["bbox"][0]
            box[1] += step["bbox"][1]
            results.append((i, 1.0 if iou(box, gt_bbox).item() > 0.5 else 0.0, n_steps))
        if args.shard == "samples":
            results = sorted(r for part in gather_objects(results, world) for r in part)
        if rank == 0:
            hits, lengths = [r[1] for r in results], [r[2] for r in results]
            print("Avg search path length:", np.mean([n for n, h in zip(lengths, hits) if h]))
            print("Top 1 Acc:", np.mean(hits))
            if args.output_path:
                import torch.distributed as tdist
                json.dump({"world_size": world, "shard": args.shard, "hits": hits, "path_lengths": lengths,
                           "backend": tdist.get_backend() if tdist.is_available() and tdist.is_initialized() else None,
                           "engine_comm": bool(getattr(vsm, "use_engine_comm", False)),
                           "rank0_search_stats": {k: v for k, v in stats.items() if k != "per_search"}},
                          open(os.path.join(args.output_path, "results.json") if args.visualization else args.output_path, "w"))
        finished = True
    finally:
        finalize(finished)


if __name__ == "__main__":
    main(sys.argv[1:])
