"""Visit frequencies behind SpeculationPolicy's priors (the priors were constants never compared with anything).

Replays best-first searches (the scheduler of vstar_amd/search.py == the reference's, tests/golden/search_paths.json) on a CPU
stand-in VSM whose per-crop outputs are random functions of (image, box, target) and whose stop probability per node is set by
`conf_shift`, and counts — over every step of every search — how often a speculation candidate of each kind was visited LATER:

    child            a child of the node being scored                      -> p_child
    queue[r]         the r-th best entry of the priority queue at that step  -> p_queue * queue_decay ** r
    queue_child[r]   a child of that entry                                   -> (its parent's probability) * p_child

Regimes are labelled by the mean number of visited nodes per search; the reference reports ~4.65 on V*Bench.
CPU only, seconds.  Prints a JSON report (committed as profiles/r04_speculation_priors.json)."""
import json
import os
import sys
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vstar_amd import search  # noqa: E402
from vstar_amd.synthetic import synthetic_image  # noqa: E402


class StandIn:
    supports_gpu_preprocess = True
    supports_deferred_mismatch = False
    max_image_slots = 64

    def __init__(self, conf_shift, max_batch=32):
        from types import SimpleNamespace
        self.cfg = SimpleNamespace(max_batch=max_batch)
        self.conf_shift, self.images, self.calls = conf_shift, {}, []

    def set_image(self, image, slot=0):
        self.images[slot] = zlib.crc32(np.asarray(image.resize((16, 16))).tobytes())

    def _one(self, key, q):
        g = torch.Generator().manual_seed(zlib.crc32(repr((key, q)).encode()) % (2 ** 31))
        low = torch.randn(12, 12, generator=g) * 9
        boxes = torch.rand(32, 4, generator=g)
        scores = torch.sigmoid(torch.randn(32, 1, generator=g) * 1.5 + self.conf_shift)
        return boxes, scores, low

    def inference_boxes(self, boxes, question, mode="detection", upsample=False, slots=None, **kw):
        qs = [question] * len(boxes) if isinstance(question, str) else list(question)
        sl = [0] * len(boxes) if slots is None else list(slots)
        self.calls.append(len(boxes))
        return [self._one((self.images[s],) + tuple(int(v) for v in b), q) for s, b, q in zip(sl, boxes, qs)]

    def inference_batch(self, *a, **k):
        raise AssertionError("boxes path expected")

    def upsample_heatmap(self, low, h, w):
        return torch.clamp(torch.nn.functional.interpolate(low[None, None], (h, w), mode="bilinear", align_corners=False)[0, 0], min=0)


def replay(conf_shift, n_images=80, seed=0):
    """-> per-kind [hits, trials] and the mean number of visited nodes."""
    rng = np.random.default_rng(seed)
    counts = {}
    log = []
    orig = search._NodeScorer.candidates

    def spy(self, prior):
        if self._last is not None:
            bbox, queue = self._last
            rec = [("child", 0, tuple(c)) for c in self._children(bbox)]
            for r, e in enumerate(sorted(queue.queue, key=lambda e: float(e.priority))[:6]):
                rec.append(("queue", r, tuple(e.item["bbox"])))
                rec += [("queue_child", r, tuple(c)) for c in self._children(e.item["bbox"])]
            log.append((id(self), rec))
        return []                                           # observe only: nothing is speculated

    search._NodeScorer.candidates = spy
    visited_n = []
    try:
        for k in range(n_images):
            w, h = int(rng.integers(1800, 4000)), int(rng.integers(1200, 2400))
            img = synthetic_image(w, h, 500 + k)
            vsm = StandIn(conf_shift)
            log.clear()
            st = {"keep_paths": True}
            search.visual_search_stream(vsm, [(img, f"thing {k}", None, search.smallest_size_for(w, h, 4.0))], window=1, stats=st,
                                        confidence_high=0.5, confidence_low=0.0, target_cue_threshold=-1.0,
                                        target_cue_threshold_minimum=-1.0)
            path = st["visit_orders"][0]
            visited_n.append(len(path))
            order = {b: i for i, b in enumerate(path)}
            for step, (_, rec) in enumerate(log):
                for kind, r, b in rec:
                    c = counts.setdefault((kind, r), [0, 0])
                    c[1] += 1
                    c[0] += 1 if order.get(b, -1) > step else 0
    finally:
        search._NodeScorer.candidates = orig
    return counts, float(np.mean(visited_n))


def main():
    report = []
    for shift in (-3.0, -3.3, -3.5, -3.6, -3.8, -4.2):
        counts, mean_nodes = replay(shift)
        freq = {f"{k}[{r}]": round(h / t, 3) for (k, r), (h, t) in sorted(counts.items()) if t >= 20}
        pq = [freq.get(f"queue[{r}]") for r in range(4)]
        decay = [round(pq[r + 1] / pq[r], 2) for r in range(3) if pq[r] and pq[r + 1] is not None]
        report.append({"conf_shift": shift, "mean_nodes_visited": round(mean_nodes, 2), "p_child": freq.get("child[0]"),
                       "p_queue_by_rank": pq, "queue_decay_by_rank": decay,
                       "p_queue_child_by_rank": [freq.get(f"queue_child[{r}]") for r in range(4)]})
        print(json.dumps(report[-1]), flush=True)
    out = os.path.join(ROOT, "profiles", "r04_speculation_priors.json")
    json.dump({"tool": "tools/calibrate_speculation.py", "regimes": report}, open(out, "w"), indent=1)
    print("->", out)


if __name__ == "__main__":
    main()
