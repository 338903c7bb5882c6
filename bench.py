#!/usr/bin/env python
"""bench.py — searched crops/sec of the VSM scoring path on MI355X (BASELINE.json metric).

One "step" = one batch of 32 crops per GPU through the whole per-crop path (CLIP-ViT-L/14@336 -> mm_projector ->
LLaMA-7B prefill over S=640 -> [LOC] hidden -> fcs heads -> OWL-ViT-B/16@768 tower -> class/box heads -> SAM-style mask
head), inputs already resident in HBM, bf16, seeded random weights of the real architecture (no network for checkpoints),
synthetic pixels/ids of the BASELINE config-2 shape.  With N GPUs each rank scores its own 32 crops (weak scaling) and
the fixed-size result records are all-gathered over RCCL every step, as the search loop needs them.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel = the bf16 MFMA GEMM,
timed live with HIP events on the engine's stream) and `cpu_baseline` (the oracle on the host cores, bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from vstar_amd import _lib  # noqa: E402
from vstar_amd.config import VSMConfig  # noqa: E402
from vstar_amd.engine import VstarEngine  # noqa: E402
from vstar_amd.weights import random_state_dict, template_chain, trained_like_state_dict  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)


def cpu_baseline(cfg: VSMConfig, text_tokens: int) -> dict:
    """The oracle (a port of the reference's per-crop graph) on the host cores: ONE crop through the real-size CLIP-L
    tower, projector, ALL 32 LLaMA-7B layers (share_layers: one layer's weights in memory, the full depth in time), heads,
    OWL-ViT tower and SAM head, fp32.  (The reference's own model_forward needs /root/reference, which does not exist on the GPU
    box; its timing in the build container — 8 cores — is quoted in DESIGN.md §5 next to this port's.)"""
    from oracle import vsm_oracle
    # torch's intra-op pool degrades badly past a few dozen threads on these shapes (256 threads: 20x slower)
    cores = min(len(os.sched_getaffinity(0)), 32)
    torch.set_num_threads(cores)
    n_l = cfg.llm_layers
    small = VSMConfig(**{**cfg.__dict__, "llm_layers": n_l, "llm_vocab": 1024})
    sd = random_state_dict(small, seed=1, dtype=torch.float32, share_layers=True)
    g = torch.Generator().manual_seed(0)
    clip = torch.randn(1, 3, cfg.clip_image_size, cfg.clip_image_size, generator=g)
    owl = torch.randn(1, 3, cfg.owl_image_size, cfg.owl_image_size, generator=g)
    L = text_tokens + 1
    ids = torch.randint(3, 1000, (1, L), generator=g)
    ids[0, 0] = 1
    ids[0, 35] = -200
    ids[0, L - 3] = 1023
    with torch.no_grad():
        t0 = time.perf_counter()
        feats = vsm_oracle.clip_features(sd, clip, cfg.clip_heads, cfg.clip_layers, cfg.clip_select_layer)
        proj = vsm_oracle._lin(sd, "model.mm_projector", feats)
        x = vsm_oracle.splice(sd, ids, proj)
        t1 = time.perf_counter()
        h = vsm_oracle.llama_prefill(sd, x, cfg.llm_heads, n_l, cfg.llm_rms_eps, cfg.llm_rope_theta)
        t2 = time.perf_counter()
        hl = h[:, -3]
        det = vsm_oracle.text_hidden_fcs(sd, "det", hl)
        seg = vsm_oracle.text_hidden_fcs(sd, "seg", hl)
        fmap = vsm_oracle.owl_visual_embs(sd, owl, cfg.owl_heads, cfg.owl_layers)
        vsm_oracle.owl_heads(sd, fmap, det.unsqueeze(1))
        vsm_oracle.sam_mask_head(sd, fmap, seg)
        t3 = time.perf_counter()
    llm = (t2 - t1) * (cfg.llm_layers / n_l)
    total = (t1 - t0) + llm + (t3 - t2)
    return {"value": round(1.0 / total, 5), "unit": "crops/s", "cores": cores, "kind": "port",
            "sample": f"1 crop fp32 on torch CPU: CLIP-L/14@{cfg.clip_image_size} + projector {t1 - t0:.2f}s, "
                      f"all {cfg.llm_layers} LLaMA-7B layers at S={x.shape[1]} {t2 - t1:.2f}s (nothing extrapolated), "
                      f"heads + OWL-ViT@768 + SAM head {t3 - t2:.2f}s.  The REFERENCE's own model_forward(inference=True) (needs "
                      "/root/reference: absent on the GPU box) on the build container's 8 cores, fp32, same crop geometry: 14.8 - 17.8 s "
                      "per crop = 0.056 - 0.068 crops/s (oracle/gen_fulldepth_golden.py logs; round 4 trained-like set @336: 38 - 39 s)"}


def _strict(args) -> bool:
    """trained_like weights decode "Sure, [LOC]." (vstar_amd.weights.template_chain): the search legs then run VSM's DEFAULT
    strict_template=True — a crop whose teacher-forced arg-max check failed would take the stepwise-decode fallback."""
    return getattr(args, "weights", "random") == "trained_like" and not getattr(args, "fake_engine", False)


def _bench_state_dict(cfg, args):
    if getattr(args, "weights", "random") == "trained_like":
        from vstar_amd.preprocess import SyntheticTokenizer
        return trained_like_state_dict(cfg, seed=0, dtype=torch.bfloat16, share_layers=True,
                                       chain=template_chain(SyntheticTokenizer(cfg.llm_vocab)))
    return random_state_dict(cfg, seed=0, dtype=torch.bfloat16, share_layers=True)


def search_leg(eng, cfg, args, rank: int, group: bool = False, geometry=None, targets: int = 0, record_paths: bool = False) -> dict:
    """BASELINE config 2 LITERALLY (SURVEY §8d "searched crops/s"): one synthetic 3840x2160 image, `--search-targets` targets,
    exhaustive depth-3 search tree (smallest_size = 540: 1 + 4 + 16 = 21 nodes per target), crops scored in 32-crop engine
    batches, with EVERYTHING the search loop does inside the timed region: image upload, GPU-side crop / pad / Pillow-exact
    resize / normalise (vstar_preprocess_crops), the engine pass, record D2H, template check, on-device heat-map statistics
    (vstar_heatmap_stats) and the scheduler's decisions (visual_search.py:390-516 semantics, vstar_amd/search.py).  The confidence
    thresholds are set so that no search stops early or enters the free-text cue branch: seeded random weights have no notion
    of 'found', and the metric counts scored crops."""
    import warnings
    from vstar_amd.preprocess import SyntheticTokenizer
    from vstar_amd.search import LazyExactPrioritize, smallest_size_for, visual_search_many
    from vstar_amd.synthetic import synthetic_image
    from vstar_amd.vsm import VSM
    # config 5 (--config5): 8K image, --minimum_size_scale 16 -> smallest_size 270 -> depth-5 tree (341 nodes per target)
    W, H = (7680, 4320) if args.config5 else (3840, 2160)
    scale = 16.0 if args.config5 else 4.0
    if geometry is not None:
        W, H, scale = geometry
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        vsm = VSM(None, engine=eng, tokenizer=SyntheticTokenizer(cfg.llm_vocab), strict_template=_strict(args))
        # each rank searches its own image (weak scaling, no collective in this leg); --rccl-selfcheck turns the product's
        # device-record all-gather on (VSM._score_sharded over the one-rank nccl group)
        vsm.shard_crops = bool(getattr(args, "rccl_selfcheck", False))
        vsm.group_prompts = group                # False: every (crop, target) pair is a full pass, like the reference's loop
        smallest = smallest_size_for(W, H, scale)
        names = [f"object {i}" for i in range(targets or args.search_targets)]
        kw = dict(confidence_high=2.0, confidence_low=0.0, target_cue_threshold=-1.0, target_cue_threshold_minimum=-1.0)
        visual_search_many(vsm, synthetic_image(W, H, 1000 + rank), names[:2], None, smallest, **kw)       # warm-up (untimed)
        for k in vsm.timers:
            vsm.timers[k] = 0
        LazyExactPrioritize.n_exact = 0
        img = synthetic_image(W, H, rank)
        torch.cuda.synchronize()
        st_many = {"keep_paths": True} if record_paths else {}
        t0 = time.perf_counter()
        res = visual_search_many(vsm, img, names, None, smallest, stats=st_many, **kw)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    crops = int(vsm.timers["crops"])
    t = vsm.timers
    depth, n, side = 1, 1, min(W, H)
    while side > smallest:
        side, depth, n = side // 2, depth + 1, n + 4 ** depth
    return {"search_crops_per_s": round(crops / dt, 2), "wall_s": round(dt, 3), "crops_scored": crops, "targets": len(names),
            "image": f"{W}x{H} synthetic", "tree": f"depth {depth} ({n} nodes per target, smallest_size {smallest}), exhaustive",
            "batch": cfg.max_batch,
            "mode": ("shared-prefix grouping of the targets that visit the same crop (vstar_vsm_score_grouped) + " if group else "") +
                    "GPU preprocessing + on-device heat-map statistics (exact float32 fallback on near-ties: "
                    f"{LazyExactPrioritize.n_exact} evaluations)",
            "stage_s": {"engine_incl_gpu_preprocess_and_record_d2h": round(t["engine_s"], 3),
                        "heatmap_statistics": round(t["post_s"], 3), "host_preprocess": round(t["preprocess_s"], 3),
                        "decisions_prompting_and_other_host": round(dt - t["engine_s"] - t["post_s"] - t["preprocess_s"] - t["gather_s"], 3)},
            "path_lengths": [int(r[1]) for r in res],
            **({"visit_orders": st_many.get("visit_orders", [])} if record_paths else {})}


class _SynthLoader:
    """A sample's image as a LOADER (what visual_search.py's loop does with Image.open): generated when the sample is about to
    enter the window — on the stream search's prefetch thread — instead of 100+ images of 25 MB held up front by every rank."""

    def __init__(self, w, h, seed):
        self.w, self.h, self.seed, self.key = w, h, seed, ("synthetic", w, h, seed)

    def __call__(self):
        from vstar_amd.synthetic import synthetic_image
        return synthetic_image(self.w, self.h, self.seed)


def stream_leg(eng, cfg, args, world: int, rank: int, shard: str = "crops", group_prompts=False, window: int = 0, samples_cap: int = 0,
               speculate: bool = True) -> dict:
    """BEST-FIRST searches that really stop: `--stream-samples` (image, target) samples — 4K synthetic
    images, `--stream-targets-per-image` targets each — searched by visual_search_stream in a window of concurrent searches
    (cross-image lock step, image slots, cost-aware speculation).  Seeded random weights have no notion of 'found', so the
    confidence threshold is CALIBRATED (untimed) to the q = 0.78 quantile of the top detection score over 128 crops: a node then
    ends its search with probability ~0.2 like a real target would, and the best-first order is exercised instead of an
    exhaustive tree.  Reported: searches/s, useful crops/s (crops the reference's order visits — the per-sample loop of
    visual_search.py:536-560 scores exactly these), engine records/s, the wasted (speculated, never visited) fraction, path lengths.
    world > 1: STRONG scaling of this fixed job — shard 'crops' (every rank walks the same searches, each engine step's crops dealt
    over the ranks, records all-gathered over RCCL: the north-star layout) or 'samples' (searches dealt over the ranks, no data-path
    collective, results gathered at the end).
    Round 4: the job is sized to the world (>= 2 x window x ranks samples, so the per-rank batches of a sharded run are not starved
    by the tail of the job); images are LOADERS generated when their sample approaches the window (prefetch thread: load + upload
    overlap the engine step); `host_serial_frac` = the share of the wall time in which no scoring step was running (uploads,
    heat-map statistics, decisions — what every rank repeats under crop sharding); group_prompts = True is the SHIPPED default of
    the VSM class (shared-prefix entry point for every locate prompt), False the plain batches whose records are bit-identical to
    the per-sample loop's; window = 1 with / without speculation is the latency regime of a lone search."""
    import warnings
    import torch.distributed as dist
    from vstar_amd.preprocess import SyntheticTokenizer
    from vstar_amd.search import smallest_size_for, visual_search_stream
    from vstar_amd.synthetic import synthetic_image
    from vstar_amd.vsm import VSM
    W, H = getattr(args, "stream_image_wh", (3840, 2160))
    on_gpu = torch.cuda.is_available() and not getattr(args, "fake_engine", False)
    red_dev = f"cuda:{eng.device}" if on_gpu else "cpu"
    sync = torch.cuda.synchronize if on_gpu else (lambda: None)
    win = window or args.stream_window or cfg.max_batch * (world if shard == "crops" else 1)
    n_samples = args.stream_samples if getattr(args, "fake_engine", False) else max(args.stream_samples, 2 * cfg.max_batch * world)
    if samples_cap:
        n_samples = min(n_samples, samples_cap)
    n_img = max(n_samples // args.stream_targets_per_image, 1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        vsm = VSM(None, engine=eng, tokenizer=SyntheticTokenizer(cfg.llm_vocab), strict_template=_strict(args))
        vsm.shard_crops = (shard == "crops") and (world > 1 or bool(getattr(args, "rccl_selfcheck", False)))
        if vsm.shard_crops and world > 1 and getattr(args, "engine_comm", "auto") != "off" and on_gpu:
            from vstar_amd.dist import maybe_engine_comm
            maybe_engine_comm(vsm)
        vsm.group_prompts = group_prompts             # False = plain batches: records bit-identical to the per-sample loop
        smallest = smallest_size_for(W, H, 4.0)
        images = [synthetic_image(W, H, 2000 + k) for k in range(min(n_img, 8))]       # calibration set; the job itself uses loaders
        loaders = [_SynthLoader(W, H, 2000 + k) for k in range(n_img)]
        samples = [(loaders[k], f"object {t}", None, smallest) for k in range(n_img) for t in range(args.stream_targets_per_image)]
        base = dict(confidence_low=0.0, target_cue_threshold=-1.0, target_cue_threshold_minimum=-1.0)
        # ---- calibration (untimed, every rank identically): top scores of the roots and of some first-level children ----
        vsm.set_image(images[0])
        boxes = [[0, 0, W, H]] + [[x, y, W // 2, H // 2] for x in (0, W // 2) for y in (0, H // 2)]
        tops = []
        sc = vsm.shard_crops
        vsm.shard_crops = False
        for k in range(min(n_img, 8)):
            vsm.set_image(images[k])
            for t in range(min(args.stream_targets_per_image, 4)):
                res = vsm.inference_boxes(boxes[:4], f"Please locate the object {t} in this image.", mode="detection", upsample=False)
                tops += [float(r[1].float().max()) for r in res]
        vsm.shard_crops = sc
        conf_high = float(np.quantile(np.asarray(tops), 0.78))
        mine = samples if shard == "crops" else samples[rank::world]
        for k in vsm.timers:
            vsm.timers[k] = 0
        if world > 1:
            dist.barrier()
        sync()
        st = {}
        t0 = time.perf_counter()
        res = visual_search_stream(vsm, mine, window=win if (window or args.stream_window) else None, stats=st, confidence_high=conf_high,
                                   speculate=speculate, **base)
        sync()
        dt = time.perf_counter() - t0
    t = dict(vsm.timers)
    if world > 1:
        # the job ends when the slowest rank ends; under 'samples' the per-rank counters add up
        tt = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        if shard == "samples":
            cnt = torch.tensor([st["searches"], st["crops_scored"], st["useful_crops"], st["engine_steps"]], dtype=torch.float64,
                               device=red_dev)
            dist.all_reduce(cnt)
            st.update(searches=int(cnt[0]), crops_scored=int(cnt[1]), useful_crops=int(cnt[2]), engine_steps=int(cnt[3]))
            st["wasted_crop_frac"] = 1.0 - st["useful_crops"] / max(st["crops_scored"], 1)
    paths = [int(r[1]) for r in res]
    visited = [int(p.get("path_visited", 0)) for p in st.get("per_search", [])]
    if getattr(args, "fake_engine", False):
        # plumbing check: a digest of every search's outcome, so that tests can compare world sizes / shard modes
        import zlib
        digest = zlib.crc32(repr([(int(r[1]), bool(r[2]), tuple(r[0]["bbox"])) for r in res]).encode())
        if world > 1 and shard == "samples":
            allp = [None] * world
            dist.all_gather_object(allp, [(int(r[1]), bool(r[2]), tuple(r[0]["bbox"])) for r in res])
            merged = [None] * len(samples)
            for rk, part in enumerate(allp):
                merged[rk::world] = part
            digest = zlib.crc32(repr(merged).encode())
        st["outcome_digest"] = digest
    steps = max(int(st["engine_steps"]), 1)
    return {"searches": int(st["searches"]), "searches_per_s": round(st["searches"] / dt, 2),
            "useful_crops_per_s": round(st["useful_crops"] / dt, 2), "records_per_s": round(st["crops_scored"] / dt, 2),
            "wasted_crop_frac": round(float(st["wasted_crop_frac"]), 4), "wall_s": round(dt, 3),
            "useful_crops": int(st["useful_crops"]), "crops_scored": int(st["crops_scored"]), "engine_steps": int(st["engine_steps"]),
            "mean_crops_per_step": round(st["crops_scored"] / steps, 2),
            "per_rank_crops_per_step": round(st["crops_scored"] / steps / (world if shard == "crops" else 1), 2),
            "mean_nodes_visited_per_search": round(float(np.mean(visited)), 2) if visited else None,
            "mean_reported_path_length": round(float(np.mean(paths)), 2),
            "window": win, "shard": shard, "ranks": world, "group_prompts": group_prompts, "speculate": speculate,
            "engine_calls": int(st.get("engine_calls", st["engine_steps"])), "async_image_uploads": int(st.get("async_uploads", 0)),
            "host_serial_frac": round(max(0.0, 1.0 - float(t["engine_s"]) / dt), 4) if dt > 0 else None,
            "images": f"{n_img} x {W}x{H} synthetic (generated by the prefetch thread as their samples approach the window), "
                      f"{args.stream_targets_per_image} targets each, depth-3 trees (smallest_size {smallest})",
            "confidence_high_calibrated": round(conf_high, 6),
            "stage_s": {"engine_incl_gpu_preprocess": round(t["engine_s"], 3), "record_allgather_and_d2h": round(t["gather_s"], 3),
                        "heatmap_statistics": round(t["post_s"], 3),
                        "allgather_us_per_step": round(t["gather_s"] / steps * 1e6, 1)},
            "order": "best-first with early stop (reference semantics); speculation by vstar_amd.search.SpeculationPolicy",
            **({"outcome_digest": st["outcome_digest"]} if "outcome_digest" in st else {})}


def small_batch_table(eng, cfg, args, dev, T: int) -> dict:
    """The latency regime each rank of an 8-way crop-sharded search lives in: the same full step at 1 / 2 / 4 / 8
    crops per rank (a 32-crop search step dealt over 8 ranks = 4 crops per rank), a few steps each."""
    import ctypes
    from vstar_amd.synthetic import bench_inputs
    vp = ctypes.c_void_p
    out = {}
    for b in (1, 2, 4, 8):
        if b > cfg.max_batch:
            break
        clip, owl, ids, loc, verify = bench_inputs(cfg, b, T, 7)
        clip, owl = clip.to(dev), owl.to(dev)
        rec = torch.empty((b, _lib.RESULT_FLOATS), dtype=torch.float32, device=dev)
        flags = _lib.F_DEVICE_INPUTS | _lib.F_DEVICE_OUTPUT

        def step():
            _lib.check(eng.lib.vstar_vsm_score_batch(eng.handle, b, vp(clip.data_ptr()), vp(owl.data_ptr()), ids.ctypes.data_as(vp), ids.shape[1],
                                                     loc.ctypes.data_as(vp), verify.ctypes.data_as(vp), verify.shape[1], flags, vp(rec.data_ptr())), eng.handle)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 6
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        out[str(b)] = {"ms_per_step": round(ms, 2), "crops_per_s": round(b / ms * 1e3, 1)}
    out["note"] = "crops per rank per step -> full-path step time on one GPU; an 8-way sharded 32-crop search step is the '4' row on every rank"
    return out


class _FakeStreamEngine:
    """CPU stand-in for VstarEngine's on-device entry points (--fake-engine): a record is a deterministic function of (image
    content, box, prompt ids).  Lets the N-process stream legs — crop sharding with the per-step record all-gather, sample sharding,
    the max-over-ranks timing — run under gloo without a GPU (tests/test_host.py).  NOT a measurement."""

    def __init__(self, cfg):
        self.cfg, self.device, self.images = cfg, 0, {}

    def set_image(self, image, slot=0):
        import zlib
        self.images[int(slot)] = zlib.crc32(np.asarray(image.resize((24, 24))).tobytes())

    def score_boxes(self, xyxy, ids, loc, verify_pos=None, raw=False, out_dev=None, share_prefix=None, slots=None):
        import zlib
        xyxy, ids = np.asarray(xyxy), np.asarray(ids)
        sl = np.zeros(len(xyxy), np.int32) if slots is None else np.asarray(slots)
        out = np.zeros((len(xyxy), _lib.RESULT_FLOATS), np.float32)
        for b in range(len(xyxy)):
            seed = zlib.crc32(repr((self.images[int(sl[b])], tuple(int(v) for v in xyxy[b]), tuple(int(t) for t in ids[b] if t))).encode())
            rng = np.random.default_rng(seed)
            out[b, :2304] = rng.standard_normal(2304) * 1.5 - 2.0
            out[b, 2304:2304 * 5] = rng.random(2304 * 4)
            out[b, 2304 * 5:2304 * 5 + 192 * 192] = np.repeat(np.repeat(rng.standard_normal((12, 12)) * 6, 16, 0), 16, 1).ravel()
        return out if raw else VstarEngine.unpack(out, 0)

    unpack = staticmethod(VstarEngine.unpack)

    def upsample_mask(self, low, h, w, clamp=True):
        t = torch.nn.functional.interpolate(torch.from_numpy(np.asarray(low, np.float32).reshape(1, 1, 192, 192)), (h, w), mode="bilinear",
                                            align_corners=False)[0, 0]
        return (t.clamp(min=0) if clamp else t).numpy()

    def heatmap_stats(self, low, h, w, rects=None):
        H = self.upsample_mask(low, h, w).astype(np.float64)
        return np.asarray([H.min(), H.max(), H.sum()] + [H[y:y + rh, x:x + rw].sum() for x, y, rw, rh in (rects or [])], np.float64)


def fake_engine_run(args, world, rank, dist):
    """The multi-process skeleton of main() with a stub in place of the engine (CPU, gloo): same collectives, same timing
    protocol, same JSON keys.  Used by tests/test_host.py to cover the N>1 launch path without GPUs."""
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    B = args.batch
    rec = torch.full((B, _lib.RESULT_FLOATS), float(rank), dtype=torch.float32)

    def step():
        time.sleep(0.01 * (1 + rank))                       # ranks finish at different times: the MAX must win
        if world > 1:
            out = torch.empty((world * B, _lib.RESULT_FLOATS), dtype=torch.float32)
            dist.all_gather_into_tensor(out, rec)
            return out
        return rec

    def fence():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        assert [float(out[r * B, 0]) for r in range(world)] == [float(r) for r in range(world)]     # rank order of the gather
    # the stream legs of main() on the stand-in engine: same driver code, gloo collectives, small images
    args.stream_image_wh = (960, 540)
    args.stream_samples, args.stream_targets_per_image = min(args.stream_samples, 12), 2
    fcfg = VSMConfig.tiny(max_batch=4, max_text_len=96)
    stream = stream_leg(_FakeStreamEngine(fcfg), fcfg, args, world, rank, "crops")
    stream_samples = stream_leg(_FakeStreamEngine(fcfg), fcfg, args, world, rank, "samples") if world > 1 else None
    if rank == 0:
        fake_line = json.dumps({"metric": "FAKE-ENGINE plumbing check (not a measurement)", "value": round(world * B * args.steps / dt, 3),
                          "unit": "crops/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "none", "data": "stub", "config": {"workload": "stub", "parallelism": f"dp{world}"},
                          "roofline": None, "cpu_baseline": None, "search_stream": stream, "search_stream_shard_samples": stream_samples})
    if world > 1:
        dist.destroy_process_group()
    if rank == 0:
        _print_last(fake_line)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="crops per GPU per step (BASELINE config 2: 32)")
    ap.add_argument("--image-size", type=int, default=336)
    ap.add_argument("--text-tokens", type=int, default=64)
    ap.add_argument("--tiny", action="store_true", help="plumbing check with the tiny-width model (NOT a valid bench)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-power-sample", action="store_true", help="skip the 2.5 s of extra steps during which rocm-smi is sampled (profiling runs)")
    ap.add_argument("--skip-owl", action="store_true", help="core path only (diagnostic; NOT the headline metric)")
    ap.add_argument("--fake-engine", action="store_true", help="CPU plumbing check of the N-process path (gloo, stub step): "
                    "exercises rank/world handling, the per-step all-gather, the barrier and the max-over-ranks timing. NOT a bench")
    ap.add_argument("--search-targets", type=int, default=16, help="targets of the config-2 search leg (>= 16 per BASELINE config 2)")
    ap.add_argument("--stream-samples", type=int, default=512, help="(image, target) samples of the best-first stream leg")
    ap.add_argument("--stream-targets-per-image", type=int, default=2)
    ap.add_argument("--stream-window", type=int, default=0, help="concurrent searches of the stream leg (0 = one engine batch x ranks)")
    ap.add_argument("--no-stream-leg", action="store_true")
    ap.add_argument("--no-latency-leg", action="store_true", help="skip the default-VSM stream run and the window-1 latency legs")
    ap.add_argument("--latency-samples", type=int, default=48, help="(image, target) samples of each window-1 latency run")
    ap.add_argument("--no-small-batch", action="store_true", help="skip the 1/2/4/8 crops-per-rank table")
    ap.add_argument("--no-config5-line", action="store_true", help="skip the bounded W8A8 (BASELINE config 5 precision) sub-object")
    ap.add_argument("--no-search-leg", action="store_true", help="skip the end-to-end search leg (N = 1 only by default)")
    ap.add_argument("--config5", action="store_true", help="BASELINE config 5 as specified: W8A8 fp8 LLaMA linears, 64-crop batches, and the "
                    "search leg on an 8K synthetic image with --minimum_size_scale 16 (depth-5 tree); separate line, not the headline")
    ap.add_argument("--rccl-selfcheck", action="store_true", help="N = 1 only: join a ONE-rank nccl (= RCCL) process group and run the N > 1 "
                    "code path — per-step all_gather_into_tensor of the device records, barrier, max-over-ranks — on the single GPU")
    ap.add_argument("--weights", choices=("random", "trained_like"), default="trained_like",
                    help="seeded synthetic weights: vstar_amd.weights.trained_like_state_dict (default since round 4: outlier channels, "
                         "massive-activation BOS, spread norm gains, peaked attention, and a greedy decode that emits the answer template, "
                         "so the search legs run the default strict_template=True path; same step time as random weights, measured) or "
                         "the i.i.d. random set of rounds 1-3")
    ap.add_argument("--engine-comm", nargs="?", const="on", default="auto", choices=["auto", "on", "off"],
                    help="N > 1: the crop-sharded stream leg gathers its records with the C-ABI's own RCCL communicator "
                    "(vstar_allgather_results on the engine stream); auto (default, round 6) = when its self-check against "
                    "torch.distributed passes on every rank, else torch.distributed; off = torch.distributed")
    ap.add_argument("--fp8", action="store_true", help="BASELINE config 5 precision: LLaMA linears W8A8 on the fp8 MFMA "
                    "(separate line; the headline metric is the default bf16 run)")
    args = ap.parse_args()

    if args.config5:
        args.fp8 = True
        if args.batch == 32:
            args.batch = 64
        if args.search_targets == 16:
            args.search_targets = 2            # 2 x 341 = 682 crops
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return _self_launch(args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks "
                         f"(use --nproc-per-node {args.gpus}, or run plain `python bench.py --gpus {args.gpus}`)")
    import torch.distributed as dist
    if args.fake_engine:
        return fake_engine_run(args, world, rank, dist)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP engine has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_group = world > 1 or args.rccl_selfcheck
    if use_group:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    B, T = args.batch, args.text_tokens
    L = T + 1
    if args.tiny:
        cfg = VSMConfig.tiny(clip_image_size=args.image_size, max_batch=B, max_text_len=L)
    else:
        cfg = VSMConfig.seal_7b(args.image_size, max_batch=B, max_text_len=L, llm_w8a8=1 if args.fp8 else 0)
    P = cfg.n_img_tokens
    S = P + T
    t0 = time.perf_counter()
    eng = VstarEngine(cfg, local_rank)
    eng.load_state_dict(_bench_state_dict(cfg, args))
    t_load = time.perf_counter() - t0

    # synthetic crop batch (vstar_amd.synthetic.bench_inputs: the same batch the full-depth reference golden was recorded on,
    # tests/test_fulldepth_gpu.py), resident in HBM before the timed region
    from vstar_amd.synthetic import bench_inputs
    clip, owl, ids, loc, verify = bench_inputs(cfg, B, T, rank)
    clip, owl = clip.to(dev), owl.to(dev)
    nv = verify.shape[1]
    rec_dev = torch.empty((B, _lib.RESULT_FLOATS), dtype=torch.float32, device=dev)
    flags = _lib.F_DEVICE_INPUTS | _lib.F_DEVICE_OUTPUT | (_lib.F_SKIP_OWL if args.skip_owl else 0)
    import ctypes
    vp = ctypes.c_void_p

    def step_local():        # this rank's scoring pass alone (no collective): what rank 0 repeats while it samples power / clock
        _lib.check(eng.lib.vstar_vsm_score_batch(
            eng.handle, B, vp(clip.data_ptr()), vp(owl.data_ptr()), ids.ctypes.data_as(vp), L, loc.ctypes.data_as(vp),
            verify.ctypes.data_as(vp), nv, flags, vp(rec_dev.data_ptr())), eng.handle)   # synchronises the engine stream

    def step():
        step_local()
        if use_group:
            out = torch.empty((world * B, _lib.RESULT_FLOATS), dtype=torch.float32, device=dev)
            dist.all_gather_into_tensor(out, rec_dev)
            return out
        return rec_dev

    def fence():
        if use_group:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if use_group:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_per_step = dt / args.steps * 1e3
    crops_per_s = world * B * args.steps / dt

    # board power and shader clock WHILE the step runs (rank 0, untimed extra steps; rocm-smi in a side thread).  Round 5 found the
    # GEMM family pinned at the board's power cap on random operands: ~1400 W with the shader clock throttled to ~1.73 GHz of its
    # 2.4 GHz (profiles/r05_power_probe.txt) — the context in which `roofline.frac` (against the 2.4-GHz peak) has to be read.
    power = None
    if rank == 0 and not args.tiny and not args.no_power_sample:
        try:
            power = _sample_power(step_local, seconds=2.5)      # rank 0 only: must not enter a collective the other ranks skip
        except Exception as exc:            # noqa: BLE001 — context only, never fatal
            power = {"error": f"{type(exc).__name__}: {exc}"}
    # roofline of the dominant kernel family (bf16 MFMA GEMM): HIP events around every GEMM launch on the engine stream
    eng.profile(True)
    for _ in range(2):
        step()
    gemm_ms, gemm_n, gemm_flops = eng.profile_read()
    f8_ms_m, f8_n_m, f8_flops_m = eng.profile_read_fp8() if args.fp8 else (0.0, 0, 0.0)
    eng.profile(False)
    achieved = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
    fl = cfg.flops_per_crop(T, full=not args.skip_owl)
    per_crop = fl["core"] if args.skip_owl else fl["full"]
    # last LLaMA block on the needed rows only (engine.hip::llm_forward): o_proj + gate/up/down of (S - 4) rows are never executed
    skipped = 2.0 * (S - 4) * cfg.llm_hidden * (cfg.llm_hidden + 3 * cfg.llm_mlp)
    # HBM-side traffic of the GEMM family per launch, from the committed rocprofv3 PMC passes of this same command
    # (FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE; profiles/r01_pmc_v7.json) — not re-measured live
    traffic, traffic_src, traffic_stale = None, None, None
    # round 6: the stamp is the hash COMPILED INTO the loaded libvstar_hip.so (build.sh -> vstar_build_source_hash), not
    # a hash of whatever sources lie in the tree when this line runs; a tree that differs from the binary is reported, not hidden
    from vstar_amd.provenance import kernel_source_hash as _tree_hash, library_source_hash as kernel_source_hash
    import glob
    # newest committed PMC summary first: rNN_pmc_final.json of the latest round, then its numbered passes (round 2 read a stale
    # file here because "_final" did not match the pattern)
    def _pmc_key(path):
        b = os.path.basename(path)
        tag = b.split("_pmc_")[-1].replace(".json", "")
        return (b[:3], 10 ** 6 if tag == "final" else int("".join(c for c in tag if c.isdigit()) or 0))
    for cand in sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_pmc_*.json")), key=_pmc_key, reverse=True):
        if "hbm" in os.path.basename(cand):
            continue
        try:
            pmc = json.load(open(cand))
            # round 5: a PMC summary is stamped with the hash of the kernel sources it was measured on; only a file of THIS build's
            # kernels is quoted (an unstamped pre-round-5 file, or another build's, leaves `traffic` null with the reason stated)
            if not isinstance(pmc, dict) or pmc.get("kernel_source_hash") != kernel_source_hash():
                traffic_stale = os.path.relpath(cand, ROOT) if traffic_stale is None else traffic_stale
                continue
            gem = [k for k in pmc["kernels"] if "gemm" in k["kernel"]]
            traffic = round(sum(k["fetch_GB_corrected"] + k["write_GB"] for k in gem) * 1e9 / sum(k["launches"] for k in gem))
            traffic_src = os.path.relpath(cand, ROOT)
            break
        except Exception:
            continue
    peak = 5000.0 if args.fp8 else PEAK_BF16_TFLOPS      # --fp8: 93 % of the GEMM FLOPs run on the fp8 MFMA (dense peak ~5 PF)
    roofline = {"bound": "mfma", "achieved": round(achieved, 1), "peak": peak, "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4), "traffic": None if args.fp8 else traffic,
                "traffic_note": (f"bytes per GEMM launch at the L2<->fabric boundary (FETCHx2+WRITE) from the committed rocprofv3 --pmc "
                                 f"passes of this command ({traffic_src}, stamped with this build's kernel_source_hash "
                                 f"{kernel_source_hash()}; counters cannot be read inside an un-profiled run); includes "
                                 "Infinity-Cache hits; algorithmic operand+output bytes per launch ~0.3 GB") if traffic is not None else
                                (f"null: no committed PMC summary carries this build's kernel_source_hash {kernel_source_hash()} "
                                 f"(newest candidate: {traffic_stale}) — re-run tools/collect_r06.sh on the GPU box"),
                "kernel_source_hash": kernel_source_hash(),
                "kernel_source_hash_matches_tree": kernel_source_hash() == _tree_hash(),
                "under_load": None if not power or "error" in power else dict(
                    power, peak_at_sclk=round(peak * power["sclk_mhz"] / 2400.0, 1) if power.get("sclk_mhz") else None,
                    frac_of_peak_at_sclk=round(achieved / (peak * power["sclk_mhz"] / 2400.0), 4) if power.get("sclk_mhz") else None,
                    note="board power / shader clock sampled with rocm-smi while the step runs; `peak` is the 2.4-GHz figure, "
                         "peak_at_sclk scales it to the clock the power cap allows on these operands"),
                "fp8_linears": None if not args.fp8 or f8_ms_m <= 0 else {
                    "achieved": round(f8_flops_m / (f8_ms_m * 1e-3) / 1e12, 1), "peak": 5000.0,
                    "frac": round(f8_flops_m / (f8_ms_m * 1e-3) / 1e12 / 5000.0, 4), "launches_per_step": int(f8_n_m) // 2},
                "kernel": "gemm4w_kernel + gemm256_kernel + gemm128_kernel (bf16 MFMA GEMM family, all epilogues; gemm4w = the LLaMA linears, 76 % of the step)", "launches_per_step": gemm_n // 2,
                "avg_launch_ms": round(gemm_ms / max(gemm_n, 1), 4),
                "gemm_share_of_step": round(gemm_ms / 2 / ms_per_step, 3),
                # whole step / peak.  ALGORITHMIC FLOPs count the last LLaMA block on every row like the reference computes it;
                # the engine runs that block on the ~4 rows per crop the heads read, so the EXECUTED figure is lower
                "end_to_end_tflops_per_gpu_algorithmic": round(crops_per_s / world * per_crop / 1e12, 1),
                "end_to_end_frac_algorithmic": round(crops_per_s / world * per_crop / 1e12 / PEAK_BF16_TFLOPS, 4),
                "skipped_flops_per_crop": round(skipped, 1),
                "end_to_end_tflops_per_gpu": round(crops_per_s / world * (per_crop - skipped) / 1e12, 1),
                "end_to_end_frac": round(crops_per_s / world * (per_crop - skipped) / 1e12 / PEAK_BF16_TFLOPS, 4)}

    # end-to-end search legs.  `search` / `search_grouped`: BASELINE config 2 literally (exhaustive depth-3 trees, 16 targets of one
    # image) — single-GPU runs.  `search_stream`: best-first searches that stop, 64 (image, target) samples in a cross-image window —
    # ALSO at N > 1, as a STRONG-scaling job in both shard modes (the weak-scaling step above stays the headline `value`).
    search = search_grouped = stream = stream_samples = None
    small = None
    if world == 1 and not args.no_search_leg and not args.skip_owl:
        try:
            search = search_leg(eng, cfg, args, rank, group=False)
        except Exception as exc:            # the headline line must survive a failure of the auxiliary leg
            search = {"error": f"{type(exc).__name__}: {exc}"}
        try:
            # the same search with the targets' shared work done once per crop (results agree to bf16 rounding): what the
            # drop-in classes do by default for multi-target searches
            search_grouped = search_leg(eng, cfg, args, rank, group=True)
        except Exception as exc:
            search_grouped = {"error": f"{type(exc).__name__}: {exc}"}
    if not args.no_stream_leg and not args.no_search_leg and not args.skip_owl and (not args.tiny or args.rccl_selfcheck):
        try:
            stream = stream_leg(eng, cfg, args, world, rank, "crops")
            if world > 1:
                stream_samples = stream_leg(eng, cfg, args, world, rank, "samples")
        except Exception as exc:
            stream = {"error": f"{type(exc).__name__}: {exc}"}
    if world == 1 and not args.no_small_batch and not args.skip_owl and not args.tiny and B >= 8:
        try:
            small = small_batch_table(eng, cfg, args, dev, T)
        except Exception as exc:
            small = {"error": f"{type(exc).__name__}: {exc}"}
    # the SHIPPED default of the VSM class (group_prompts = True -> every locate prompt through the shared-prefix entry point) under
    # the stream driver, and the latency regime where the speculation policy acts (a lone search at a time: window 1), with and
    # without speculation — N = 1 only, bounded sample counts
    stream_default = latency = None
    if world == 1 and stream is not None and "error" not in stream and not args.no_latency_leg:
        try:
            stream_default = stream_leg(eng, cfg, args, world, rank, "crops", group_prompts=True)
        except Exception as exc:
            stream_default = {"error": f"{type(exc).__name__}: {exc}"}
        try:
            lat = {}
            for name, spec in (("speculate", True), ("no_speculation", False)):
                r = stream_leg(eng, cfg, args, world, rank, "crops", window=1, samples_cap=args.latency_samples, speculate=spec)
                lat[name] = {"ms_per_search": round(r["wall_s"] / max(r["searches"], 1) * 1e3, 1), "searches": r["searches"],
                             "crops_scored": r["crops_scored"], "crops_visited": r["useful_crops"],
                             "wasted_crop_frac": r["wasted_crop_frac"], "engine_steps": r["engine_steps"],
                             "mean_crops_per_step": r["mean_crops_per_step"], "wall_s": r["wall_s"]}
            lat["speedup_from_speculation"] = round(lat["no_speculation"]["ms_per_search"] / max(lat["speculate"]["ms_per_search"], 1e-9), 3)
            lat["note"] = ("window 1: one (image, target) search at a time, the reference's schedule (visual_search.py:536-560); "
                           "SpeculationPolicy scores likely-next crops in the same engine step")
            latency = lat
        except Exception as exc:
            latency = {"error": f"{type(exc).__name__}: {exc}"}
    # BASELINE config 5 (fp8 W8A8 LLaMA linears, 64-crop batches, 8K image, depth-5 tree), bounded to a few seconds inside the default
    # command: the batch path at 64 crops with its own roofline (fp8 MFMA GEMMs timed by HIP events, peak 5 PF), the literal search
    # leg on ONE target (341 crops), and the same search on the bf16 engine as the decision yardstick (visit order, final node)
    config5 = None
    if world == 1 and not args.fp8 and not args.no_config5_line and not args.skip_owl and not args.tiny:
        try:
            B8 = 64
            cfg8 = VSMConfig.seal_7b(args.image_size, max_batch=B8, max_text_len=L, llm_w8a8=1)
            eng8 = VstarEngine(cfg8, local_rank)
            eng8.load_state_dict(_bench_state_dict(cfg8, args))
            clip8, owl8, ids8, loc8, verify8 = bench_inputs(cfg8, B8, T, rank)
            clip8, owl8 = clip8.to(dev), owl8.to(dev)
            rec8 = torch.empty((B8, _lib.RESULT_FLOATS), dtype=torch.float32, device=dev)

            def step8():
                _lib.check(eng8.lib.vstar_vsm_score_batch(
                    eng8.handle, B8, vp(clip8.data_ptr()), vp(owl8.data_ptr()), ids8.ctypes.data_as(vp), L, loc8.ctypes.data_as(vp),
                    verify8.ctypes.data_as(vp), nv, flags, vp(rec8.data_ptr())), eng8.handle)
            for _ in range(2):
                step8()
            torch.cuda.synchronize()
            t8 = time.perf_counter()
            for _ in range(3):
                step8()
            torch.cuda.synchronize()
            ms8 = (time.perf_counter() - t8) / 3 * 1e3
            eng8.profile(True)
            step8()
            g8_ms, g8_n, g8_flops = eng8.profile_read()
            f8_ms, f8_n, f8_flops = eng8.profile_read_fp8()          # the launches that ran on the fp8 MFMA, apart
            eng8.profile(False)
            # decision-level agreement with the bf16 engine on the first 32 crops of this batch (the reference has no fp8 path: the
            # bf16 engine, pinned to the reference, is the yardstick)
            same_box = None
            if B <= B8:
                step()                                                  # the bf16 engine's records of the headline batch ...
                _lib.check(eng8.lib.vstar_vsm_score_batch(             # ... and the W8A8 engine's records of the SAME crops
                    eng8.handle, B, vp(clip.data_ptr()), vp(owl.data_ptr()), ids.ctypes.data_as(vp), L, loc.ctypes.data_as(vp),
                    verify.ctypes.data_as(vp), nv, flags, vp(rec8.data_ptr())), eng8.handle)
                a, b8 = rec_dev.cpu().numpy(), rec8[:B].cpu().numpy()
                same_box = float(np.mean(a[:, :2304].argmax(1) == b8[:, :2304].argmax(1)))
            geo = (7680, 4320, 16.0)
            s8 = search_leg(eng8, cfg8, args, rank, group=False, geometry=geo, targets=1, record_paths=True)
            s16 = search_leg(eng, cfg, args, rank, group=False, geometry=geo, targets=1, record_paths=True)
            vo8, vo16 = s8.pop("visit_orders", [[]])[0], s16.pop("visit_orders", [[]])[0]
            common = 0
            for x, y in zip(vo8, vo16):
                if x != y:
                    break
                common += 1
            config5 = {"workload": "BASELINE config 5: W8A8 fp8 LLaMA linears, 64-crop batches; search leg = one 7680x4320 synthetic image, "
                                   "--minimum_size_scale 16 (depth-5 tree, 341 nodes), 1 target",
                       "crops_per_s": round(B8 / ms8 * 1e3, 2), "ms_per_step": round(ms8, 2), "steps": 3, "batch": B8,
                       "dtype": "fp8 e4m3 W8A8 LLaMA linears on v_mfma_scale_f32_16x16x128_f8f6f4, bf16 elsewhere",
                       "roofline": {"bound": "mfma", "achieved": round(g8_flops / (g8_ms * 1e-3) / 1e12, 1) if g8_ms > 0 else None,
                                    "peak": 5000.0, "unit": "TFLOP/s",
                                    "frac": round(g8_flops / (g8_ms * 1e-3) / 1e12 / 5000.0, 4) if g8_ms > 0 else None,
                                    "note": "all GEMM launches of one 64-crop step timed by HIP events (93 % of their FLOPs run on the "
                                            "fp8 MFMA; the ViT / head GEMMs stay bf16), against the dense fp8 peak",
                                    # round 5: the two populations apart, each against its own peak
                                    "fp8_linears": {"launches": int(f8_n), "achieved": round(f8_flops / (f8_ms * 1e-3) / 1e12, 1) if f8_ms > 0 else None,
                                                    "peak": 5000.0, "frac": round(f8_flops / (f8_ms * 1e-3) / 1e12 / 5000.0, 4) if f8_ms > 0 else None,
                                                    "share_of_gemm_time": round(f8_ms / g8_ms, 3) if g8_ms > 0 else None},
                                    "bf16_gemms": {"launches": int(g8_n - f8_n),
                                                   "achieved": round((g8_flops - f8_flops) / ((g8_ms - f8_ms) * 1e-3) / 1e12, 1) if g8_ms > f8_ms else None,
                                                   "peak": PEAK_BF16_TFLOPS,
                                                   "frac": round((g8_flops - f8_flops) / ((g8_ms - f8_ms) * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4) if g8_ms > f8_ms else None}},
                       "argmax_box_same_as_bf16_engine": None if same_box is None else round(same_box, 4),
                       "search": {k: s8[k] for k in ("search_crops_per_s", "wall_s", "crops_scored", "tree", "stage_s")},
                       "search_bf16_engine_crops_per_s": s16["search_crops_per_s"],
                       "same_visit_order": bool(vo8 == vo16 and len(vo8) > 0),
                       "common_visit_prefix": common, "nodes_visited": [len(vo8), len(vo16)],
                       "same_final_box_and_path_length": s8["path_lengths"] == s16["path_lengths"],
                       "weights": args.weights}
            eng8.close()
        except Exception as exc:
            config5 = {"error": f"{type(exc).__name__}: {exc}"}

    if rank == 0:
        cpu = None
        if not args.no_cpu_baseline and not args.tiny:
            # rank 0 times the CPU port at every N (the other ranks wait in the closing barrier, outside every timed region): a
            # SCALE line without it would count as unmeasured
            cpu = cpu_baseline(cfg, T)
            if world > 1:
                cpu["note"] = f"timed on rank 0 while the other {world - 1} ranks idle in the closing barrier"
        line = {
            "metric": "searched crops/sec (336x336 tiles, 7B VSM " + ("W8A8 fp8" if args.fp8 else "bf16") + ")",
            "value": round(crops_per_s, 3), "unit": "crops/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "fp8 e4m3 W8A8 for the LLaMA linears (per-token / per-channel scales), bf16 elsewhere" if args.fp8 else "bf16",
            "data": "synthetic (seeded " + ("trained-like" if args.weights == "trained_like" else "random") +
                    " weights of the real architecture, N(0,1) pixels, random ids)",
            "config": {"workload": ("TINY-plumbing " if args.tiny else "") + ("[config-5 precision] " if args.fp8 else "") +
                       f"BASELINE config {5 if args.config5 else 2}: {B}-crop batches/GPU, CLIP-ViT-L/14@{cfg.clip_image_size} (P={P}) + LLaMA-7B prefill "
                       f"S={S} + " + ("(core only)" if args.skip_owl else "OWL-ViT-B/16@768 + det/SAM heads") +
                       ", records all-gathered per step",
                       "crops_per_gpu_per_step": B, "text_tokens": T, "seq_len": S, "parallelism": f"dp{world}",
                       "flops_per_crop": per_crop, "weights_load_s": round(t_load, 1)},
            "roofline": roofline, "cpu_baseline": cpu, "search": search, "search_grouped": search_grouped,
            "search_stream": stream, "search_stream_shard_samples": stream_samples, "search_stream_default_vsm": stream_default,
            "search_latency": latency, "per_rank_shape": small, "config5": config5,
            "world_size": world, "collective": None if not use_group else {
                "backend": dist.get_backend() + " (RCCL over xGMI)", "op": "all_gather_into_tensor of the per-crop result records, once per step",
                "bytes_per_rank_per_step": int(B * _lib.RESULT_FLOATS * 4), "ranks": dist.get_world_size()},
        }
    if use_group:
        _flush_c_stdio()                 # every rank: whatever RCCL buffered through C stdio goes out BEFORE rank 0's JSON line
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        _print_last(json.dumps(line))


def _sample_power(step, seconds: float = 2.5) -> dict:
    """Runs `step` for ~`seconds` (untimed) while a side thread polls `rocm-smi --showpower --showclocks --json`; returns the mean
    board power (W) and shader clock (MHz) over the samples taken under load."""
    import re
    import subprocess
    import threading

    def sample():
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=10).stdout
        card = next(iter(json.loads(out).values()))
        pw = next((float(v) for k, v in card.items() if "ower" in k and re.match(r"^[\d.]+$", str(v))), None)
        ck = next((str(v) for k, v in card.items() if "sclk" in k.lower() and "speed" in k.lower()), "")
        m = re.search(r"(\d+)\s*Mhz", ck, re.I)
        return pw, float(m.group(1)) if m else None

    samples, stop = [], threading.Event()

    def loop():
        while not stop.is_set():
            try:
                samples.append(sample())
            except Exception:            # noqa: BLE001
                pass
            time.sleep(0.25)
    th = threading.Thread(target=loop, daemon=True)
    th.start()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        step()
    torch.cuda.synchronize()
    stop.set()
    th.join(timeout=15)
    pw = [s[0] for s in samples[1:] if s[0] is not None]
    ck = [s[1] for s in samples[1:] if s[1] is not None]
    if not pw and not ck:
        return {"error": "rocm-smi returned no power / clock fields"}
    return {"power_w": round(sum(pw) / len(pw), 0) if pw else None, "sclk_mhz": round(sum(ck) / len(ck), 0) if ck else None, "samples": len(samples)}


def _self_launch(n: int) -> None:
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset): start the N ranks ourselves — the same
    `torch.distributed.run --nnodes=1 --nproc-per-node N` command the driver uses, rendezvous on 127.0.0.1 and a free port — and
    pass its exit code on.  The children inherit stdout, so rank 0's JSON line is still the last line (every rank flushes its C
    stdio and passes a barrier before rank 0 prints)."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    raise SystemExit(subprocess.call(cmd, env=env))


def _print_last(text: str) -> None:
    """The JSON line must be the LAST line of stdout: RCCL writes its start-up banner through C stdio, which (on a pipe) stays
    buffered until exit and would otherwise land after it.  Flush the C buffers first, then print and flush."""
    _flush_c_stdio()
    print(text, flush=True)


def _flush_c_stdio() -> None:
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


if __name__ == "__main__":
    main()
