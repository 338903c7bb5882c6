"""The batched scheduler reproduces the REFERENCE scheduler's decisions (tests/golden/search_paths.json was recorded
by running the reference's own visual_search.py with oracle.search_oracle.FakeVSM, see oracle/gen_search_golden.py)."""
import json
import os
import types

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from vstar_amd.synthetic import synthetic_image
from oracle.search_oracle import FakeVSM
from vstar_amd import search

ALL_GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "search_paths.json")))
GOLD = [g for g in ALL_GOLD if not g.get("cue")]
CUE_GOLD = [g for g in ALL_GOLD if g.get("cue")]


class BatchedFake(FakeVSM):
    """FakeVSM behind the batched interface of vstar_amd.vsm.VSM (low-res result + deferred upsample)."""

    def __init__(self, max_batch, **kw):
        super().__init__(**kw)
        self.cfg = types.SimpleNamespace(max_batch=max_batch)
        self.batches = []

    def inference_batch(self, images, question, mode="detection", upsample=False):
        self.batches.append(len(images))
        out = []
        for im in images:
            self.calls += 1
            g = self._rng(im)
            low = self._low(g, self.gain)
            boxes = torch.rand(self.n_boxes, 4, generator=g)
            scores = torch.sigmoid(torch.randn(self.n_boxes, 1, generator=g) * 1.5 + self.conf_shift)
            out.append((boxes, scores, low[0, 0]))
        return out

    def upsample_heatmap(self, low, h, w):
        return torch.clamp(F.interpolate(low[None, None], (h, w), mode="bilinear", align_corners=False)[0, 0], min=0)

    def heatmap_stats(self, low, h, w, rects=None):
        """What vstar_heatmap_stats returns, computed in float64 from the same up-sampled map."""
        H = self.upsample_heatmap(low, h, w).double().numpy()
        out = [H.min(), H.max(), H.sum()]
        for x, y, rw, rh in (rects or []):
            out.append(H[y:y + rh, x:x + rw].sum())
        return np.asarray(out, np.float64)


def _check(res, gold):
    final_step, path_length, ok, all_valid = res
    assert int(path_length) == gold["path_length"]
    assert bool(ok) == gold["success"]
    assert [int(v) for v in final_step["bbox"]] == gold["final_bbox"]
    assert [float(v) for v in final_step["detection_result"]] == gold["detection_result"]   # bit-identical
    assert (None if all_valid is None else int(all_valid.shape[0])) == gold["n_all_valid"]


@pytest.mark.parametrize("gold", GOLD, ids=[f"{g['case'][0]}x{g['case'][1]}s{g['case'][2]}" for g in GOLD])
def test_unbatched_matches_reference(gold):
    w, h, iseed, vseed, shift, scale = gold["case"]
    img = synthetic_image(w, h, iseed)
    smallest = search.smallest_size_for(w, h, scale)
    assert smallest == gold["smallest_size"]
    vsm = FakeVSM(seed=vseed, conf_shift=shift, symmetric=gold.get("symmetric", False))
    _check(search.visual_search(vsm, img, "object", [0, 0, 10, 10], smallest), gold)
    assert vsm.calls == gold["calls"]          # no speculation without a batch interface


@pytest.mark.parametrize("max_batch", [4, 32])
@pytest.mark.parametrize("gold", GOLD, ids=[f"{g['case'][0]}x{g['case'][1]}s{g['case'][2]}" for g in GOLD])
def test_batched_speculative_matches_reference(gold, max_batch):
    w, h, iseed, vseed, shift, scale = gold["case"]
    img = synthetic_image(w, h, iseed)
    vsm = BatchedFake(max_batch, seed=vseed, conf_shift=shift, symmetric=gold.get("symmetric", False))
    stats = {}
    _check(search.visual_search(vsm, img, "object", [0, 0, 10, 10], gold["smallest_size"], stats=stats, device_reductions=False), gold)
    assert max(vsm.batches) <= max_batch
    assert stats["crops_scored"] >= min(gold["calls"], stats["path_visited"])
    # speculation reduces engine passes: never more batches than the reference made single-crop calls
    assert stats["engine_batches"] <= gold["calls"]
    if gold["calls"] >= 21 and max_batch == 32:
        assert stats["engine_batches"] <= (gold["calls"] + 15) // 16


@pytest.mark.parametrize("gold", GOLD, ids=[f"{g['case'][0]}x{g['case'][1]}s{g['case'][2]}" for g in GOLD])
def test_device_reductions_path_matches_reference(gold):
    """The algebraic form of the decision math used with on-device heat-map statistics (min-max normalisation folded into
    rectangle sums, fp64) takes the same decisions as the reference's float32 numpy reductions."""
    w, h, iseed, vseed, shift, scale = gold["case"]
    img = synthetic_image(w, h, iseed)
    sym = gold.get("symmetric", False)
    vsm = BatchedFake(32, seed=vseed, conf_shift=shift, symmetric=sym)
    dev, host = {}, {}
    search.LazyExactPrioritize.n_exact = 0
    _check(search.visual_search(vsm, img, "object", [0, 0, 10, 10], gold["smallest_size"], stats=dev), gold)      # default = device path
    n_exact = search.LazyExactPrioritize.n_exact
    search.visual_search(BatchedFake(32, seed=vseed, conf_shift=shift, symmetric=sym), img, "object", [0, 0, 10, 10],
                         gold["smallest_size"], stats=host, device_reductions=False)
    # the whole VISIT ORDER equals the float32 host path's (which is the reference's arithmetic), not just the final box
    assert [p["bbox"] for p in dev["search_path"]] == [p["bbox"] for p in host["search_path"]]
    if sym:
        # mirror-symmetric heat maps: sibling scores tie up to float32 rounding, so the order is the reference's only because
        # near-tied queue entries fall back to the exact float32 reduction (LazyExactPrioritize) — check that it really engaged
        assert n_exact > 0
    elif max(len(dev["search_path"]), 1) > 1:
        assert n_exact <= 2 * len(dev["search_path"])          # and that ordinary searches almost never pay for it


class SparseFake(BatchedFake):
    """Heat maps like a trained segmentation head's: negative (clamped to zero) on the background, positive in one small blob — so
    most children of a node carry EXACTLY zero mass, in the reference's float32 arithmetic and in the device statistics alike."""

    def _low(self, g, scale):
        low = torch.full((1, 1, 12, 12), -3.0)
        i, j = (int(v) for v in torch.randint(0, 6, (2,), generator=g))
        low[0, 0, i, j] = float(torch.rand(1, generator=g)) * scale + 1.0
        return low


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_exact_zero_children_tie_without_the_float32_fallback(seed):
    """Round 4: children whose heat mass is exactly zero (every pixel of the clamped map inside them is 0) tie EXACTLY in the
    reference's arithmetic too, so ordering them must not materialise heat maps (160 of 336 queue pushes of the config-2 search leg
    did on trained-like weights).  The visit order still equals the float32 host path's."""
    img = synthetic_image(1920, 1080, 10 + seed)
    kw = dict(confidence_high=2.0, confidence_low=0.0, target_cue_threshold=-1.0, target_cue_threshold_minimum=-1.0)
    dev, host = {}, {}
    search.LazyExactPrioritize.n_exact = 0
    search.visual_search(SparseFake(32, seed=seed, conf_shift=-6.0), img, "o", None, 270, stats=dev, **kw)
    n_exact = search.LazyExactPrioritize.n_exact
    search.visual_search(SparseFake(32, seed=seed, conf_shift=-6.0), img, "o", None, 270, stats=host, device_reductions=False, **kw)
    assert [p["bbox"] for p in dev["search_path"]] == [p["bbox"] for p in host["search_path"]]
    assert len(dev["search_path"]) == 21
    zeros = sum(1 for p in dev["search_path"][1:] if float(p["score"]) == 0.0)
    assert zeros >= 10                                     # the regime is real: most children carry no mass at all
    assert n_exact == 0, n_exact


def test_device_reductions_child_scores_close_to_numpy_path():
    img = synthetic_image(1920, 1080, 1)
    a, b = {}, {}
    kw = dict(confidence_high=2.0)
    search.visual_search(BatchedFake(32, seed=1, conf_shift=-6.0), img, "o", None, 270, stats=a, **kw)
    search.visual_search(BatchedFake(32, seed=1, conf_shift=-6.0), img, "o", None, 270, stats=b, device_reductions=True, **kw)
    sa = [float(p["score"]) for p in a["search_path"][1:]]
    sb = [float(p["score"]) for p in b["search_path"][1:]]
    assert len(sa) == len(sb) == 20
    assert np.allclose(sa, sb, rtol=1e-5, atol=1e-7)


def test_helpers_match_reference_semantics():
    assert search.split_4subpatches([0, 0, 100, 200]) == (1, 4)
    assert search.split_4subpatches([0, 0, 200, 100]) == (4, 1)
    assert search.split_4subpatches([0, 0, 100, 150]) == (2, 2)
    subs, ws, hs = search.get_sub_patches([10, 20, 101, 77], 2, 2)
    assert subs == [[10, 20, 50, 38], [60, 20, 51, 38], [10, 58, 50, 39], [60, 58, 51, 39]] and (ws, hs) == (50, 38)
    assert search.smallest_size_for(3840, 2160) == 540 and search.smallest_size_for(600, 400) == 224
    z = np.zeros((4, 4, 1), np.float32)
    assert [float(s) for s in search.get_subpatch_scores(z, [0, 0, 4, 4], [[0, 0, 2, 2]])] == [0.0]
    assert abs(search.iou([0, 0, 10, 10], [5, 5, 10, 10]) - 25 / 175) < 1e-12


@pytest.mark.parametrize("gold", CUE_GOLD, ids=[f"cue{g['case'][0]}x{g['case'][1]}" for g in CUE_GOLD])
def test_contextual_cue_branch_matches_reference(gold):
    """score_max <= threshold: VQA text -> location phrase -> segmentation heatmap (visual_search.py:427-443), recorded from
    the reference with spaCy stubbed to an empty parse (=> "region <phrase>")."""
    from oracle.gen_search_golden import CUE_TEXT
    w, h, iseed, vseed, shift, scale = gold["case"]
    img = synthetic_image(w, h, iseed)
    vsm = FakeVSM(seed=vseed, conf_shift=shift, gain=0.1, vqa_text=CUE_TEXT)
    stats = {}
    _check(search.visual_search(vsm, img, "object", [0, 0, 10, 10], gold["smallest_size"], noun_chunker=lambda s: [],
                                stats=stats), gold)
    assert vsm.calls == gold["calls"]
    assert sum(1 for m, _ in vsm.questions if m == "vqa") == gold["n_vqa"]
    assert sorted({q for m, q in vsm.questions if m == "segmentation"}) == gold["seg_questions"]
    assert stats["search_path"][0]["context_cue"].startswith(CUE_TEXT + "#region")


def test_contextual_cue_branch_needs_vqa():
    # heat.max() <= threshold sends the search into the VQA branch, which the round-1 engine does not implement
    img = synthetic_image(1920, 1080, 0)
    vsm = FakeVSM(seed=1, conf_shift=-6.0, gain=0.1)
    with pytest.raises(NotImplementedError):
        search.visual_search(vsm, img, "object", [0, 0, 1, 1], 270)


class _MismatchFake(BatchedFake):
    """BatchedFake whose crops smaller than `min_side` fail the template check: with defer_mismatch the slot holds a
    DeferredMismatch that raises the reference's IndexError only when resolved."""
    supports_deferred_mismatch = True

    def __init__(self, min_side, **kw):
        super().__init__(**kw)
        self.min_side = min_side
        self.resolved = 0

    def inference_batch(self, images, question, mode="detection", upsample=False, defer_mismatch=False):
        from vstar_amd.vsm import DeferredMismatch
        out = super().inference_batch(images, question, mode, upsample)

        def boom():
            self.resolved += 1
            raise IndexError("index -1 is out of bounds for dimension 0 with size 0")
        for i, im in enumerate(images):
            if min(im.size) < self.min_side:
                assert defer_mismatch, "the scheduler must ask for deferred mismatches"
                out[i] = DeferredMismatch(boom)
        return out


def test_speculative_template_mismatch_only_raises_when_visited():
    """Round-1 advisor finding: a template mismatch on a speculative crop the best-first order never visits must not abort a
    search the reference would complete; on a crop the search DOES visit, the reference's IndexError surfaces."""
    # case 8: confident detection on the whole image (path_length 1) — children are speculated but never visited
    gold = [g for g in GOLD if g["case"][2] == 8][0]
    w, h, iseed, vseed, shift, scale = gold["case"]
    img = synthetic_image(w, h, iseed)
    vsm = _MismatchFake(min(w, h), max_batch=32, seed=vseed, conf_shift=shift)
    stats = {}
    _check(search.visual_search(vsm, img, "object", [0, 0, 10, 10], gold["smallest_size"], stats=stats), gold)
    assert stats["crops_scored"] > 1 and vsm.resolved == 0          # speculated, mismatching, never consumed
    # case 1: low confidence everywhere, the search descends — the first visited child raises like the reference would
    gold = [g for g in GOLD if g["case"][2] == 1][0]
    w, h, iseed, vseed, shift, scale = gold["case"]
    img = synthetic_image(w, h, iseed)
    vsm = _MismatchFake(min(w, h), max_batch=32, seed=vseed, conf_shift=shift)
    with pytest.raises(IndexError):
        search.visual_search(vsm, img, "object", [0, 0, 10, 10], gold["smallest_size"])
    assert vsm.resolved == 1


# ---------------- lock-step multi-target search (visual_search_many): thread machinery on a CPU stand-in ----------------
class _BoxVSM:
    """On-device-style VSM stand-in: crops travel as boxes (`inference_boxes`), results are a deterministic function of
    (box, question); low-res heat maps are upsampled on demand.  Records every engine call."""
    supports_gpu_preprocess = True
    supports_deferred_mismatch = False

    def __init__(self, max_batch=6, fail_on=None):
        from types import SimpleNamespace
        self.cfg = SimpleNamespace(max_batch=max_batch)
        self.calls, self.fail_on = [], fail_on
        self.image = None

    def set_image(self, image):
        self.image = image

    def _one(self, b, q):
        import zlib
        g = torch.Generator().manual_seed(zlib.crc32(repr((tuple(int(v) for v in b), q)).encode()) % (2 ** 31))
        low = torch.randn(12, 12, generator=g) * 9
        boxes = torch.rand(32, 4, generator=g)
        scores = torch.sigmoid(torch.randn(32, 1, generator=g) * 1.5 - 2.5)
        return boxes, scores, low

    def inference_boxes(self, boxes, question, mode="detection", upsample=False, **kw):
        qs = [question] * len(boxes) if isinstance(question, str) else list(question)
        self.calls.append([(tuple(int(v) for v in b), q) for b, q in zip(boxes, qs)])
        if self.fail_on is not None and any(self.fail_on in q for q in qs):
            raise RuntimeError("engine failure")
        return [self._one(b, q) for b, q in zip(boxes, qs)]

    def inference_batch(self, *a, **k):            # presence marks the VSM as batched
        raise AssertionError("boxes path expected")

    def upsample_heatmap(self, low, h, w):
        return torch.clamp(torch.nn.functional.interpolate(low[None, None], (h, w), mode="bilinear", align_corners=False)[0, 0], min=0)


def test_lock_step_many_equals_the_per_target_loop():
    from vstar_amd.search import smallest_size_for, visual_search, visual_search_many
    from vstar_amd.synthetic import synthetic_image
    img = synthetic_image(960, 640, 3)
    smallest = smallest_size_for(960, 640, 4.0)
    names = ["kite", "red umbrella", "dog", "traffic light", "boat"]
    kw = dict(confidence_high=0.97, confidence_low=0.0, target_cue_threshold=-1.0, target_cue_threshold_minimum=-1.0, batch_size=3)
    solo_vsm = _BoxVSM()
    loop = [visual_search(solo_vsm, img, n, None, smallest, **kw) for n in names]
    many_vsm = _BoxVSM()
    many = visual_search_many(many_vsm, img, names, None, smallest, **kw)
    assert len(many) == len(loop)
    for a, b in zip(loop, many):
        assert a[1] == b[1] and a[2] == b[2] and a[0]["bbox"] == b[0]["bbox"]
        assert float(a[0]["score"] or 0) == float(b[0]["score"] or 0)
    # lock step: fewer, fuller engine calls than the loop, and every call holds the requests of several targets while they last
    assert len(many_vsm.calls) < len(solo_vsm.calls)
    assert len({q for _, q in many_vsm.calls[0]}) == len(names)
    # the same (box, question) pairs were scored in total
    flat = lambda calls: sorted(p for c in calls for p in c)  # noqa: E731
    assert flat(many_vsm.calls) == flat(solo_vsm.calls)
    # target i's result does not depend on the company: a different set of co-searched targets gives the same tuple
    other = visual_search_many(_BoxVSM(), img, [names[2], "zebra"], None, smallest, **kw)
    assert other[0][1] == loop[2][1] and other[0][0]["bbox"] == loop[2][0]["bbox"]


def test_lock_step_engine_failure_reaches_the_caller():
    from vstar_amd.search import smallest_size_for, visual_search_many
    from vstar_amd.synthetic import synthetic_image
    img = synthetic_image(640, 480, 1)
    with pytest.raises(RuntimeError, match="engine failure"):
        visual_search_many(_BoxVSM(fail_on="dog"), img, ["kite", "dog", "boat"], None, smallest_size_for(640, 480, 4.0),
                           confidence_high=2.0, confidence_low=0.0, target_cue_threshold=-1.0, target_cue_threshold_minimum=-1.0)


@pytest.mark.parametrize("seed", range(6))
def test_lock_step_many_equals_loop_on_random_settings(seed):
    """Image sizes, batch sizes, target counts and thresholds drawn at random: lock-step results == the per-target loop, and the
    engine scored the same (crop, question) pairs at most once each."""
    from vstar_amd.search import smallest_size_for, visual_search, visual_search_many
    rng = np.random.default_rng(100 + seed)
    w, h = int(rng.integers(500, 1400)), int(rng.integers(400, 1000))
    img = synthetic_image(w, h, seed)
    smallest = smallest_size_for(w, h, float(rng.choice([2.0, 4.0, 8.0])), 64)
    names = [f"thing {k}" for k in range(int(rng.integers(2, 6)))]
    kw = dict(confidence_high=float(rng.choice([0.6, 0.9, 2.0])), confidence_low=float(rng.choice([0.0, 0.3])),
              target_cue_threshold=-1.0, target_cue_threshold_minimum=-1.0, batch_size=int(rng.integers(1, 9)),
              speculate=bool(rng.integers(0, 2)))
    solo, many_vsm = _BoxVSM(max_batch=4), _BoxVSM(max_batch=4)
    loop = [visual_search(solo, img, n, None, smallest, **kw) for n in names]
    many = visual_search_many(many_vsm, img, names, None, smallest, **kw)
    for a, b in zip(loop, many):
        assert a[1] == b[1] and a[2] == b[2] and a[0]["bbox"] == b[0]["bbox"]
        assert torch.equal(a[0]["detection_result"], b[0]["detection_result"])
    pairs = [p for c in many_vsm.calls for p in c]
    assert len(pairs) == len(set(pairs))
    assert sorted(pairs) == sorted(p for c in solo.calls for p in c)


class _BoxVSMStats(_BoxVSM):
    """_BoxVSM + on-device-style heat-map statistics (single and batched), recording how the scheduler asks for them."""

    def __init__(self, **kw):
        super().__init__(**kw)
        self.stat_calls = []

    @property
    def supports_device_reductions(self):
        return True

    def heatmap_stats(self, low, h, w, rects=None):
        self.stat_calls.append(1)
        return self._stats(low, h, w, rects)

    def _stats(self, low, h, w, rects):
        H = self.upsample_heatmap(torch.as_tensor(np.asarray(low)), h, w).double().numpy()
        return np.asarray([H.min(), H.max(), H.sum()] + [H[y:y + rh, x:x + rw].sum() for x, y, rw, rh in (rects or [])], np.float64)

    def heatmap_stats_batch(self, items):
        self.stat_calls.append(len(items))
        return [self._stats(*it) for it in items]


def test_lock_step_batches_the_heat_map_statistics_of_all_targets():
    """Device reductions in lock step: every target's statistics requests of a round (a node's own map and its ancestors') are
    served by ONE batched call; the searches equal the per-target loop's, which equal the one-call-per-map path."""
    from vstar_amd.search import smallest_size_for, visual_search, visual_search_many
    img = synthetic_image(1100, 700, 5)
    smallest = smallest_size_for(1100, 700, 4.0)
    names = ["kite", "dog", "boat"]
    kw = dict(confidence_high=2.0, confidence_low=0.0, target_cue_threshold=-1.0, target_cue_threshold_minimum=-1.0)
    loop_vsm = _BoxVSMStats()
    loop = [visual_search(loop_vsm, img, n, None, smallest, **kw) for n in names]
    many_vsm = _BoxVSMStats()
    many = visual_search_many(many_vsm, img, names, None, smallest, **kw)
    for a, b in zip(loop, many):
        assert a[1] == b[1] and a[2] == b[2] and a[0]["bbox"] == b[0]["bbox"]
        assert float(a[0]["score"] or 0) == float(b[0]["score"] or 0)
    assert sum(many_vsm.stat_calls) == sum(loop_vsm.stat_calls)          # the same statistics in total ...
    assert len(many_vsm.stat_calls) < len(loop_vsm.stat_calls)           # ... in fewer calls
    assert max(many_vsm.stat_calls) >= len(names)                        # a round's requests of all targets in one call


# ---------------- cross-image lock-step search (visual_search_stream): window of concurrent samples, image slots, cost-aware
# speculation, useful-work statistics ----------------
class _SlotVSM(_BoxVSMStats):
    """_BoxVSM with image SLOTS: a record is a function of (image content, box, question); every call is logged with its slots."""
    max_image_slots = 64

    def __init__(self, **kw):
        super().__init__(**kw)
        self.images, self.uploads = {}, []

    def set_image(self, image, slot=0):
        import zlib
        self.images[slot] = zlib.crc32(np.asarray(image.resize((16, 16))).tobytes())
        self.uploads.append(slot)

    def inference_boxes(self, boxes, question, mode="detection", upsample=False, slots=None, **kw):
        qs = [question] * len(boxes) if isinstance(question, str) else list(question)
        sl = [0] * len(boxes) if slots is None else list(slots)
        assert len(boxes) <= self.cfg.max_batch * 2          # (+ the do-not-split-a-crop overhang)
        self.calls.append([(self.images[s], tuple(int(v) for v in b), q) for s, b, q in zip(sl, boxes, qs)])
        return [self._one((self.images[s],) + tuple(int(v) for v in b), q) for s, b, q in zip(sl, boxes, qs)]


def _stream_samples(n_images=5, per_image=(1, 2, 1, 3, 1), seed=0):
    rng = np.random.default_rng(seed)
    samples = []
    for k in range(n_images):
        w, h = int(rng.integers(600, 1300)), int(rng.integers(500, 900))
        img = synthetic_image(w, h, 50 + k)
        for t in range(per_image[k % len(per_image)]):
            samples.append((img, f"thing {k}-{t}", None, search.smallest_size_for(w, h, 4.0, 64)))
    return samples


@pytest.mark.parametrize("window,conf", [(1, 0.9), (3, 0.9), (8, 0.9), (8, 2.0), (4, 0.6)])
def test_stream_equals_the_per_sample_loop(window, conf):
    """visual_search_stream over (image, target) samples of SEVERAL images == visual_search per sample, whatever the window;
    engine batches mix crops of different images; statistics account for every scored crop."""
    samples = _stream_samples()
    kw = dict(confidence_high=conf, confidence_low=0.0, target_cue_threshold=-1.0, target_cue_threshold_minimum=-1.0)
    loop_vsm = _SlotVSM(max_batch=8)
    loop = [search.visual_search(loop_vsm, img, n, gt, sm, speculate=False, **kw) for img, n, gt, sm in samples]
    vsm = _SlotVSM(max_batch=8)
    st = {}
    got = search.visual_search_stream(vsm, samples, window=window, stats=st, **kw)
    assert len(got) == len(samples)
    for a, b in zip(loop, got):
        assert a[1] == b[1] and a[2] == b[2] and a[0]["bbox"] == b[0]["bbox"]
        assert torch.equal(a[0]["detection_result"], b[0]["detection_result"])
    # useful work = what the reference's order visits = what the speculation-free loop scored
    useful = sum(len(c) for c in loop_vsm.calls)
    assert st["useful_crops"] == useful and st["searches"] == len(samples)
    assert st["crops_scored"] == sum(len(c) for c in vsm.calls) >= useful
    assert abs(st["wasted_crop_frac"] - (1 - useful / st["crops_scored"])) < 1e-12
    if window >= 3:
        assert any(len({p[0] for p in c}) > 1 for c in vsm.calls)         # crops of different images in one engine call
        assert len(vsm.calls) < len(loop_vsm.calls)
    # one upload per distinct image that was live (samples sharing an image object share the slot while they overlap)
    assert len(vsm.uploads) <= len(samples)


def test_stream_recycles_image_slots_and_keeps_sample_order():
    samples = _stream_samples(n_images=7, per_image=(1,))
    kw = dict(confidence_high=0.9, confidence_low=0.0, target_cue_threshold=-1.0, target_cue_threshold_minimum=-1.0)
    loop = [search.visual_search(_SlotVSM(max_batch=8), img, n, gt, sm, speculate=False, **kw) for img, n, gt, sm in samples]
    vsm = _SlotVSM(max_batch=8)
    vsm.max_image_slots = 2                          # fewer slots than the window asks for: searches wait for a free slot
    got = search.visual_search_stream(vsm, samples, window=6, **kw)
    assert set(vsm.uploads) == {0, 1} and len(vsm.uploads) == 7
    for a, b in zip(loop, got):
        assert a[1] == b[1] and a[0]["bbox"] == b[0]["bbox"]


def test_speculation_policy_cost_model():
    pol = search.SpeculationPolicy({1: 20.0, 2: 26.0, 4: 40.0, 8: 68.0, 16: 126.0, 32: 236.0}, cap=32)
    assert pol.step_ms(1) == 20.0 and pol.step_ms(3) == 33.0 and pol.step_ms(32) == 236.0 and pol.step_ms(64) == 472.0
    cands = [(0.45, "c0"), (0.45, "c1"), (0.45, "c2"), (0.45, "c3"), (0.5, "q0"), (0.25, "q1"), (0.1, "far")]
    # a lone search (latency mode): p * t(1) = 9 ms >= marginal 6-7 ms for the likely candidates, never for the unlikely ones —
    # and (round 4) never more of them than one single-crop step's worth of extra time: t(4) - t(1) = 20 ms = t(1), t(5) - t(1) = 27
    lone = pol.select(1, cands, 1)
    assert lone == ["q0", "c0", "c1"]
    # many live searches: a hit rarely shortens the schedule -> nothing is worth a crop's marginal cost
    assert pol.select(8, cands, 8) == []
    assert pol.select(2, cands, 2) == []
    # the cap bounds a step even for certain candidates
    assert len(pol.select(30, [(1.0, k) for k in range(10)], 1)) == 2
    # data-parallel ranks: B crops cost a step of ceil(B / world) crops per rank
    pol8 = search.SpeculationPolicy({1: 20.0, 2: 26.0, 4: 40.0}, cap=256, world=8)
    assert pol8.step_ms(8) == 20.0 and pol8.step_ms(9) == 26.0
    assert len(pol8.select(1, [(0.25, k) for k in range(20)], 1)) == 7       # free until the ranks hold one crop each
    assert search.SpeculationPolicy(enabled=False).select(1, cands, 1) == []


def test_stream_speculates_for_a_lone_search_but_not_in_a_full_window():
    samples = _stream_samples(n_images=6, per_image=(1,))
    kw = dict(confidence_high=2.0, confidence_low=0.0, target_cue_threshold=-1.0, target_cue_threshold_minimum=-1.0)
    st1, st6 = {}, {}
    v1 = _SlotVSM(max_batch=8)
    search.visual_search_stream(v1, samples[:1], window=1, stats=st1, **kw)
    assert max(len(c) for c in v1.calls) > 1                 # latency mode: children ride along with the node
    v6 = _SlotVSM(max_batch=8)
    search.visual_search_stream(v6, samples, window=6, stats=st6, **kw)
    assert st6["wasted_crop_frac"] == 0.0                    # six live searches: only crops the order visits
    assert st6["useful_crops"] == st6["crops_scored"]


@pytest.mark.parametrize("conf", [0.5, 0.7, 0.9, 2.0])
def test_speculation_never_costs_a_lone_search_more_than_one_step(conf):
    """in the regime the policy exists for (window 1 — one search at a time) the modelled time of every search —
    sum of t(B) over its engine steps with the MI355X step table — must not exceed the no-speculation schedule's by more than one
    step, and the results must be the same.  Priors: measured visit frequencies (profiles/r04_speculation_priors.json)."""
    samples = _stream_samples(n_images=8, per_image=(1, 2))
    kw = dict(confidence_high=conf, confidence_low=0.0, target_cue_threshold=-1.0, target_cue_threshold_minimum=-1.0)
    pol = search.SpeculationPolicy(cap=8)
    lose, total_spec, total_plain = 0.0, 0.0, 0.0
    for smp in samples:
        va, vb = _SlotVSM(max_batch=8), _SlotVSM(max_batch=8)
        ra = search.visual_search_stream(va, [smp], window=1, policy=pol, **kw)          # one policy: it learns the regime as it goes
        rb = search.visual_search_stream(vb, [smp], window=1, speculate=False, **kw)
        assert ra[0][1] == rb[0][1] and ra[0][2] == rb[0][2] and ra[0][0]["bbox"] == rb[0][0]["bbox"]
        ta = sum(pol.step_ms(len(c)) for c in va.calls)
        tb = sum(pol.step_ms(len(c)) for c in vb.calls)
        assert all(len(c) == 1 for c in vb.calls)
        assert ta <= tb + pol.step_ms(1) + 1e-9, (ta, tb)
        lose = max(lose, ta - tb)
        total_spec += ta
        total_plain += tb
    assert total_spec <= 1.05 * total_plain + 2 * pol.step_ms(1)      # over the set: at worst a wash (plus what it cost to learn the regime)


def test_speculation_priors_follow_the_observed_hit_rate():
    """Closed loop (round 4): the policy learns from the searches that finished how many of its speculative crops were really
    visited, and scales every candidate's probability by observed / predicted hits.  Exhaustive searches (everything speculated is
    visited sooner or later) push the calibration above 1, searches that mostly end at their second node push it below 1 — and a
    policy with explicit priors is left alone."""
    samples = _stream_samples(n_images=8, per_image=(1, 2))
    base = dict(confidence_low=0.0, target_cue_threshold=-1.0, target_cue_threshold_minimum=-1.0)
    up, down = search.SpeculationPolicy(cap=8), search.SpeculationPolicy(cap=8)
    for smp in samples:
        search.visual_search_stream(_SlotVSM(max_batch=8), [smp], window=1, policy=up, confidence_high=2.0, **base)
        search.visual_search_stream(_SlotVSM(max_batch=8), [smp], window=1, policy=down, confidence_high=0.6, **base)
    assert up.calibration > 1.05 and up._pred > 0 and up._hits > up._pred
    assert down.calibration < 0.95
    fixed = search.SpeculationPolicy(cap=8, p_child=0.4)
    search.visual_search_stream(_SlotVSM(max_batch=8), samples[:3], window=1, policy=fixed, confidence_high=2.0, **base)
    assert fixed.calibration == 1.0 and fixed.p_child == 0.4
    # a low calibration makes the policy pickier: fewer candidates pass the cost test
    cands = [(0.75, "q0"), (0.525, "q1"), (0.37, "q2"), (0.25, "c0")]
    fresh = search.SpeculationPolicy(cap=8)
    n_before = len(fresh.select(1, cands, 1))
    fresh.calibration = 0.5
    assert len(fresh.select(1, cands, 1)) < n_before


@pytest.mark.parametrize("prefetch", [0, 1, 3])
def test_stream_prefetches_image_loaders_without_changing_anything(prefetch):
    """Lazy loaders (what visual_search.py / vstar_bench_eval.py hand over: open + decode a file) run ahead of the window on a worker
    thread when prefetch > 0: each exactly once, results and upload order identical to loading on entry."""
    import threading
    base = _stream_samples(n_images=6, per_image=(1, 2))
    calls, threads = [], set()

    class Loader:
        def __init__(self, k, img):
            self.key, self.img = ("file", k), img

        def __call__(self):
            calls.append(self.key)
            threads.add(threading.current_thread().name)
            return self.img

    by_img, samples = {}, []
    for img, name, gt, sm in base:
        ld = by_img.setdefault(id(img), Loader(len(by_img), img))
        samples.append((ld, name, gt, sm))
    kw = dict(confidence_high=0.9, confidence_low=0.0, target_cue_threshold=-1.0, target_cue_threshold_minimum=-1.0)
    want = search.visual_search_stream(_SlotVSM(max_batch=8), base, window=3, **kw)
    vsm = _SlotVSM(max_batch=8)
    vsm.max_image_slots = 3
    got = search.visual_search_stream(vsm, samples, window=3, prefetch=prefetch, **kw)
    for a, b in zip(want, got):
        assert a[1] == b[1] and a[2] == b[2] and a[0]["bbox"] == b[0]["bbox"]
    assert sorted(calls) == sorted(ld.key for ld in by_img.values())          # every file opened once
    if prefetch:
        assert any(t.startswith("vstar-image-prefetch") for t in threads)      # ... and not all of them on the search thread
    else:
        assert threads == {threading.current_thread().name}


class _AsyncSlotVSM(_SlotVSM):
    """_SlotVSM with the asynchronous upload of the real VSM (set_image_async from the prefetch threads): records who uploaded which
    slot from which thread, and fails if a crop is scored from a slot whose upload has not been issued."""
    supports_async_upload = True

    def __init__(self, **kw):
        import threading
        super().__init__(**kw)
        self.async_uploads, self.lock = [], threading.Lock()

    def set_image_async(self, image, slot):
        import threading
        import zlib
        with self.lock:
            self.images[slot] = zlib.crc32(np.asarray(image.resize((16, 16))).tobytes())
            self.async_uploads.append((slot, threading.current_thread().name))

    def release_image(self, slot=0):
        pass


@pytest.mark.parametrize("slow_ms,window", [(0, 3), (30, 3), (30, 8)])
def test_stream_async_uploads_from_prefetch_threads_change_nothing(slow_ms, window):
    """Round 4: with an engine that offers asynchronous uploads the prefetch threads load AND upload the next samples' images into
    image slots reserved ahead; in one process a sample whose image is not ready joins a later step (slow loaders: the first steps run
    while the first window is still loading).  Results equal the synchronous path's; every image is uploaded once per residency, the
    prefetched ones from a worker thread; no crop is ever cut from a slot before its upload was issued (the stand-in would raise)."""
    import threading
    import time
    base = _stream_samples(n_images=7, per_image=(1, 2))
    calls = []

    class Loader:
        def __init__(self, k, img):
            self.key, self.img = ("file", k), img

        def __call__(self):
            time.sleep(slow_ms / 1e3)
            calls.append(self.key)
            return self.img

    by_img, samples = {}, []
    for img, name, gt, sm in base:
        ld = by_img.setdefault(id(img), Loader(len(by_img), img))
        samples.append((ld, name, gt, sm))
    kw = dict(confidence_high=0.9, confidence_low=0.0, target_cue_threshold=-1.0, target_cue_threshold_minimum=-1.0)
    want = search.visual_search_stream(_SlotVSM(max_batch=8), base, window=window, **kw)
    vsm = _AsyncSlotVSM(max_batch=8)
    st = {}
    got = search.visual_search_stream(vsm, samples, window=window, stats=st, **kw)
    for a, b in zip(want, got):
        assert a[1] == b[1] and a[2] == b[2] and a[0]["bbox"] == b[0]["bbox"]
    assert sorted(calls) == sorted(ld.key for ld in by_img.values())                      # every file opened once
    assert len(vsm.async_uploads) + len(vsm.uploads) == len(by_img)                        # ... and uploaded once
    assert st["async_uploads"] == len(vsm.async_uploads) >= 1
    assert all(t.startswith("vstar-image-prefetch") for _, t in vsm.async_uploads)
    assert threading.active_count() < 12                                                    # the pool is shut down
    # explicit opt-out: everything on the search thread again
    v2 = _AsyncSlotVSM(max_batch=8)
    search.visual_search_stream(v2, samples, window=window, async_upload=False, **kw)
    assert v2.async_uploads == [] and len(v2.uploads) == len(by_img)


def test_stream_prefetch_propagates_loader_errors():
    base = _stream_samples(n_images=3, per_image=(1,))

    def broken():
        raise OSError("cannot identify image file")

    samples = [(lambda img=img: img, n, gt, sm) for img, n, gt, sm in base]
    samples[2] = (broken, "thing", None, 64)
    kw = dict(confidence_high=0.9, confidence_low=0.0, target_cue_threshold=-1.0, target_cue_threshold_minimum=-1.0)
    with pytest.raises(OSError):
        search.visual_search_stream(_SlotVSM(max_batch=8), samples, window=1, prefetch=2, **kw)
