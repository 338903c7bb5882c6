"""The drop-in VSM class and the batched search loop on the MI355X (tiny-width model, synthetic weights/tokenizer):
return conventions of visual_search.py:208-225, parity with the oracle on the SAME preprocessed inputs, batch == single,
and batched-speculative search == one-crop-at-a-time search."""
import warnings

import numpy as np
import pytest
import torch

from oracle import vsm_oracle
from vstar_amd.synthetic import synthetic_image
from vstar_amd import preprocess as pp
from vstar_amd.config import VSMConfig
from vstar_amd.engine import VstarEngine
from vstar_amd.search import smallest_size_for, visual_search, visual_search_many
from vstar_amd.vsm import VSM
from vstar_amd.weights import template_chain, trained_like_state_dict

pytestmark = pytest.mark.gpu


def random_state_dict(cfg, seed, dtype):
    """Round 4: the module's weights have a trained checkpoint's statistics AND answer locate prompts with "Sure, [LOC]." under
    greedy decoding (vstar_amd.weights.trained_like_state_dict), so every test below runs the DEFAULT strict_template=True path:
    a crop whose teacher-forced arg-max check failed would take the stepwise-decode fallback and appear in vsm.fallback_log."""
    return trained_like_state_dict(cfg, seed=seed, dtype=dtype, share_layers=False,
                                   chain=template_chain(pp.SyntheticTokenizer(cfg.llm_vocab)))


@pytest.fixture(scope="module")
def vsm(cuda):
    cfg = VSMConfig.tiny(max_batch=8, max_text_len=96)
    eng = VstarEngine(cfg, 0)
    eng.load_state_dict(random_state_dict(cfg, seed=5, dtype=torch.bfloat16))
    with warnings.catch_warnings():
        warnings.simplefilter("error")                  # a "tolerated template mismatch" warning would fail the module
        v = VSM(None, engine=eng, tokenizer=pp.SyntheticTokenizer(cfg.llm_vocab), strict_template=True)
    yield v
    assert v.fallback_log == [], v.fallback_log[:2]     # no crop of any test needed the fallback: the template was always emitted


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def test_inference_conventions_and_oracle(vsm):
    img = synthetic_image(500, 300, 3)
    q = pp.LOCATE_QUESTION.format("blue kite")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        boxes, scores, heat = vsm.inference(img, q, mode="detection")
        seg = vsm.inference(img, q, mode="segmentation")
    assert boxes.shape == (2304, 4) and scores.shape == (2304, 1) and heat.shape == (300, 500)
    assert vsm.strict_template and vsm.last_template_ok.all()
    assert heat.dtype == torch.float32 and float(heat.min()) >= 0 and 0 < float(scores.min()) and float(scores.max()) < 1
    assert scores.dtype == torch.bfloat16 and boxes.dtype == torch.float32
    assert torch.equal(seg, heat)
    # oracle on the identical preprocessed tensors / ids
    cfg = vsm.cfg
    ids, loc_pos, _, _ = vsm._ids(q)
    clip = torch.from_numpy(pp.clip_preprocess(img, 224)).bfloat16()[None]
    owl = torch.from_numpy(pp.owl_preprocess(img, 768)).bfloat16()[None]
    sd = {k: v.float() for k, v in random_state_dict(cfg, seed=5, dtype=torch.bfloat16).items()}
    ref = vsm_oracle.vsm_forward(sd, cfg, clip.float(), owl.float(), torch.from_numpy(ids.astype(np.int64))[None], vsm.loc_token_idx)
    assert int(ref["loc_pos"][0]) == loc_pos
    # the same algorithm in bf16 on torch-CPU is the noise yardstick (tests/_parity.py): no fixed floors
    from _parity import assert_mask_within_bf16_noise, assert_within_bf16_noise
    sd16 = random_state_dict(cfg, seed=5, dtype=torch.bfloat16)
    r16 = vsm_oracle.vsm_forward(sd16, cfg, clip, owl, torch.from_numpy(ids.astype(np.int64))[None], vsm.loc_token_idx)
    box_noise = float((r16["pred_boxes"][0].float() - ref["pred_boxes"][0]).abs().max())
    assert np.abs(boxes.numpy() - ref["pred_boxes"][0].numpy()).max() < max(1e-2, 1.5 * box_noise)
    assert_within_bf16_noise("boxes", boxes.numpy(), ref["pred_boxes"][0].numpy(), r16["pred_boxes"][0].float().numpy())
    assert_within_bf16_noise("scores", scores.float().numpy(), torch.sigmoid(ref["pred_logits"][0]).numpy(),
                             torch.sigmoid(r16["pred_logits"][0]).float().numpy())
    low = vsm.inference_batch([img], q, mode="segmentation", upsample=False)[0].numpy()
    assert_mask_within_bf16_noise(low, ref["low_res_masks"][0, 0].numpy(), r16["low_res_masks"][0, 0].float().numpy(),
                                  ref["sam_taps"]["sam_hyper"].numpy(), r16["sam_taps"]["sam_hyper"].float().numpy(),
                                  ref["sam_taps"]["sam_c2"].mean(dim=1).numpy(), r16["sam_taps"]["sam_c2"].float().mean(dim=1).numpy())
    # the full-resolution heat map is exactly the engine's bilinear upsample + clamp of that low-res mask
    assert np.array_equal(heat.numpy(), vsm.engine.upsample_mask(low, 300, 500))


def test_batch_equals_single(vsm):
    imgs = [synthetic_image(300 + 40 * i, 260, 10 + i) for i in range(5)]
    q = pp.LOCATE_QUESTION.format("dog")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        batch = vsm.inference_batch(imgs, q, mode="detection")
        for i in (0, 3):
            b, s, h = vsm.inference(imgs[i], q, mode="detection")
            assert torch.equal(b, batch[i][0]) and torch.equal(s, batch[i][1]) and torch.equal(h, batch[i][2])


def test_batched_search_equals_sequential(vsm):
    img = synthetic_image(1280, 720, 21)
    smallest = smallest_size_for(1280, 720)
    kw = dict(confidence_high=2.0, confidence_low=0.0, target_cue_threshold=-1.0, target_cue_threshold_minimum=-1.0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        s_seq, s_bat = {}, {}
        r_seq = visual_search(vsm, img, "kite", None, smallest, speculate=False, stats=s_seq, **kw)
        r_bat = visual_search(vsm, img, "kite", None, smallest, speculate=True, stats=s_bat, **kw)
    assert r_seq[1] == r_bat[1] and r_seq[2] == r_bat[2]
    assert r_seq[0]["bbox"] == r_bat[0]["bbox"]
    assert torch.equal(r_seq[0]["detection_result"], r_bat[0]["detection_result"])
    assert [p["bbox"] for p in s_seq["search_path"]] == [p["bbox"] for p in s_bat["search_path"]]
    # exhaustive search (confidence_high=2 is unreachable): 1 + 4 + 16 nodes for 1280x720 with smallest_size 224
    assert s_seq["path_visited"] == 21 and s_seq["engine_batches"] == 21
    assert s_bat["engine_batches"] <= 4 and s_bat["crops_scored"] == 21


def test_multi_target_search_equals_per_target_loop(vsm):
    """visual_search_many (the targets' searches advance in lock step; prompts of different lengths right-padded in the same
    batch) returns exactly what the reference's per-object loop returns."""
    img = synthetic_image(1280, 720, 33)
    smallest = smallest_size_for(1280, 720)
    names = ["kite", "small red umbrella on the beach", "dog"]          # different prompt lengths
    kw = dict(confidence_high=2.0, confidence_low=0.0, target_cue_threshold=-1.0, target_cue_threshold_minimum=-1.0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        loop, st_loop = [], []
        for n in names:
            st = {}
            loop.append(visual_search(vsm, img, n, None, smallest, stats=st, **kw))
            st_loop.append(st)
        st_many = [{} for _ in names]
        vsm.group_prompts = False          # plain batches (different prompts right-padded in one batch): bit-identical to the loop
        try:
            many = visual_search_many(vsm, img, names, None, smallest, **kw)
        finally:
            vsm.group_prompts = True
    assert len(many) == len(loop)
    for a, b in zip(loop, many):
        assert a[1] == b[1] and a[2] == b[2] and a[0]["bbox"] == b[0]["bbox"]
        assert torch.equal(a[0]["detection_result"], b[0]["detection_result"])          # bit-identical scores
        assert float(a[0]["score"] if a[0]["score"] is not None else 0) == float(b[0]["score"] if b[0]["score"] is not None else 0)


def test_multi_target_search_with_shared_prefix_grouping(vsm):
    """visual_search_many with prompt grouping (default): the crops shared by the targets go through the towers and the shared
    LLaMA positions once (vstar_vsm_score_grouped).  Records are a second bf16 evaluation of the same numbers (different attention
    tiling), so they agree with the plain path to bf16 noise; on these fixtures every search still takes the same path."""
    img = synthetic_image(1280, 720, 33)
    smallest = smallest_size_for(1280, 720)
    names = ["kite", "small red umbrella on the beach", "dog", "traffic light"]
    kw = dict(confidence_high=2.0, confidence_low=0.0, target_cue_threshold=-1.0, target_cue_threshold_minimum=-1.0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        vsm.timers["grouped_records"] = 0
        grouped = visual_search_many(vsm, img, names, None, smallest, **kw)
        n_grouped = vsm.timers["grouped_records"]
        vsm.group_prompts = False
        try:
            plain = visual_search_many(vsm, img, names, None, smallest, **kw)
        finally:
            vsm.group_prompts = True
    # lock step: EVERY record of every target went through the grouped entry point (group_prompts = "always" during the call)
    assert n_grouped == 21 * len(names)
    # a target's result does not depend on its company: searched with other partners it takes the same path to the same numbers
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        other = visual_search_many(vsm, img, [names[2], "zebra crossing", names[0]], None, smallest, **kw)
    for got, want in ((other[0], grouped[2]), (other[2], grouped[0])):
        assert got[1] == want[1] and got[2] == want[2] and got[0]["bbox"] == want[0]["bbox"]
        assert torch.equal(got[0]["detection_result"], want[0]["detection_result"])          # bit-identical records
    for a, b in zip(plain, grouped):
        assert a[1] == b[1] and a[2] == b[2] and a[0]["bbox"] == b[0]["bbox"]
        assert torch.allclose(a[0]["detection_result"], b[0]["detection_result"], atol=1.0)      # pixels; bf16-level box differences
    # per-record agreement on one crop: scores of the two paths within bf16 noise of each other
    vsm.set_image(img)
    qs = [pp.LOCATE_QUESTION.format(n) for n in names]
    box = [[0, 0, 1280, 720]] * len(names)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        g = vsm.inference_boxes(box, qs, mode="detection", upsample=False)
        vsm.group_prompts = False
        try:
            p_ = vsm.inference_boxes(box, qs, mode="detection", upsample=False)
        finally:
            vsm.group_prompts = True
    for (bg, sg, hg), (bp, sp, hp) in zip(g, p_):
        assert torch.equal(bg, bp) or (bg - bp).abs().max() < 8e-3     # box head: identical inputs, once per crop vs once per pair
        assert (sg.float() - sp.float()).abs().max() < 2e-2
        assert rel_l2(hg.numpy() - hg.numpy().mean(), hp.numpy() - hp.numpy().mean()) < 6e-2


@pytest.mark.parametrize("use_cache", [True, False])
def test_vqa_mode_greedy_decode_matches_oracle(vsm, use_cache):
    """mode='vqa': KV-cached decode (vstar_vsm_generate) and the reference's literal no-cache greedy loop — every
    engine-chosen token must be the oracle's arg-max given the same prefix (or within bf16 noise of it when the top-2 logits
    are nearly tied)."""
    img = synthetic_image(400, 300, 8)
    q = pp.CUE_QUESTION.format("kite")
    new = vsm.generate_ids(img, q, max_new_tokens=6, use_cache=use_cache)
    assert 1 <= len(new) <= 6
    text = vsm.inference(img, q, mode="vqa")
    assert isinstance(text, str)
    cfg = vsm.cfg
    sd = {k: v.float() for k, v in random_state_dict(cfg, seed=5, dtype=torch.bfloat16).items()}
    clip = torch.from_numpy(pp.clip_preprocess(img, 224)).bfloat16().float()[None]
    ids = pp.tokenizer_image_token(pp.build_prompt(q), vsm.vsm_tokenizer)
    for tok in new:
        logits = vsm_oracle.greedy_next_logits(sd, cfg, clip, torch.tensor([ids]))[0]
        span = float(logits.max() - logits.min())
        assert float(logits.max() - logits[tok]) <= 0.02 * span, (tok, int(logits.argmax()))
        ids.append(tok)


def test_cached_decode_equals_no_cache_decode(vsm):
    """Same tokens from the KV-cached path and from re-prefilling the whole prefix per token (the reference's schedule),
    compared while the no-cache path's own decision margin is not a near-tie."""
    img = synthetic_image(520, 380, 21)
    q = pp.CUE_QUESTION.format("red umbrella")
    a = vsm.generate_ids(img, q, max_new_tokens=10, use_cache=True)
    b = vsm.generate_ids(img, q, max_new_tokens=10, use_cache=False)
    assert len(a) >= 1 and len(b) >= 1
    n = min(len(a), len(b))
    same = sum(1 for x, y in zip(a[:n], b[:n]) if x == y)
    print("cached", a, "no-cache", b)
    assert a[0] == b[0] and same >= n - 2      # a flipped near-tie changes the suffix; synthetic logits are nearly flat


def test_gpu_preprocess_is_bit_identical_to_pil_hf_path(vsm):
    """vstar_preprocess_crops (crop + top-left pad + Pillow-exact bicubic + HF normalise, on the GPU) produces exactly the
    bf16 tensors that the host path (PIL + preprocess.py, itself bit-identical to the HF processors) feeds the engine."""
    img = synthetic_image(1400, 900, 17)
    # noisy high-frequency content exercises the antialiasing window
    arr = np.asarray(img).copy()
    arr ^= np.random.default_rng(1).integers(0, 64, size=arr.shape, dtype=np.uint8)
    from PIL import Image
    img = Image.fromarray(arr)
    vsm.set_image(img)
    boxes = [[0, 0, 1400, 900], [700, 450, 700, 450], [13, 27, 301, 555], [1000, 100, 400, 224], [5, 5, 224, 224],
             [100, 100, 768, 768], [0, 0, 150, 120]]
    xyxy = [[int(x), int(y), int(x + w), int(y + h)] for x, y, w, h in boxes]
    clip_gpu, owl_gpu = vsm.engine.preprocess_only(xyxy)
    for i, b in enumerate(xyxy):
        crop = img.crop(tuple(b))
        ref_c = torch.from_numpy(pp.clip_preprocess(crop, vsm.cfg.clip_image_size)).bfloat16().float().numpy()
        ref_o = torch.from_numpy(pp.owl_preprocess(crop, 768)).bfloat16().float().numpy()
        assert np.array_equal(clip_gpu[i], ref_c), ("clip", i, np.abs(clip_gpu[i] - ref_c).max())
        assert np.array_equal(owl_gpu[i], ref_o), ("owl", i, np.abs(owl_gpu[i] - ref_o).max())


def test_search_with_gpu_preprocess_equals_host_preprocess(vsm):
    img = synthetic_image(1280, 720, 33)
    smallest = smallest_size_for(1280, 720)
    kw = dict(confidence_high=2.0, confidence_low=0.0, target_cue_threshold=-1.0, target_cue_threshold_minimum=-1.0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a, b = {}, {}
        r_host = visual_search(vsm, img, "kite", None, smallest, stats=a, gpu_preprocess=False, **kw)
        r_gpu = visual_search(vsm, img, "kite", None, smallest, stats=b, gpu_preprocess=True, **kw)
    assert r_host[1] == r_gpu[1] and r_host[0]["bbox"] == r_gpu[0]["bbox"]
    assert torch.equal(r_host[0]["detection_result"], r_gpu[0]["detection_result"])
    assert [p["bbox"] for p in a["search_path"]] == [p["bbox"] for p in b["search_path"]]
    for pa, pb in zip(a["search_path"], b["search_path"]):
        if "final_heatmap" in pa:
            assert np.array_equal(pa["final_heatmap"], pb["final_heatmap"])


def test_heatmap_stats_kernel_matches_host_reductions(vsm):
    g = torch.Generator().manual_seed(3)
    low = (torch.randn(192, 192, generator=g) * 4).numpy()
    for (h, w) in [(540, 960), (2160, 3840), (224, 301)]:
        rects = [[0, 0, w // 2, h // 2], [w // 2, 0, w - w // 2, h // 2], [0, h // 2, w // 2, h - h // 2], [w // 3, h // 5, 17, 9]]
        st = vsm.heatmap_stats(low, h, w, rects)
        H = vsm.upsample_heatmap(low, h, w).double().numpy()
        assert st[0] == H.min() and st[1] == H.max()                  # exact: same interpolation code on the GPU
        assert abs(st[2] - H.sum()) <= 1e-9 * H.sum()
        for k, (x, y, rw, rh) in enumerate(rects):
            assert abs(st[3 + k] - H[y:y + rh, x:x + rw].sum()) <= 1e-9 * max(H.sum(), 1.0)


def test_heatmap_stats_batch_equals_single_calls(vsm):
    """vstar_heatmap_stats_batch: n maps of different output sizes / rectangle counts in one engine call == n single calls (the
    same kernels; fp64 sums agree to the order of the atomics)."""
    g = torch.Generator().manual_seed(8)
    items = []
    for k, (h, w) in enumerate([(540, 960), (2160, 3840), (224, 301), (97, 1200), (700, 700)]):
        low = (torch.randn(192, 192, generator=g) * (2 + k)).numpy()
        rects = [[0, 0, w // 2, h // 2], [w // 2, 0, w - w // 2, h // 2], [0, h // 2, w // 2, h - h // 2], [w // 3, h // 5, 17, 9]][:k]
        items.append((low, h, w, rects or None))
    got = vsm.heatmap_stats_batch(items)
    assert len(got) == len(items)
    for (low, h, w, rects), st in zip(items, got):
        one = vsm.heatmap_stats(low, h, w, rects)
        assert st.shape == one.shape and st[0] == one[0] and st[1] == one[1]
        assert np.allclose(st[2:], one[2:], rtol=1e-12, atol=0)
    assert vsm.heatmap_stats_batch([]) == []


def test_heatmap_stats_survives_image_regrow(vsm):
    """Regression (round-1 advisor finding): vstar_image_set used to hipFree the heat-map statistics scratch when a LARGER image
    forced a regrow and kept the dangling pointer — the next vstar_heatmap_stats then DMA'd into freed memory that the new image
    allocation could own.  Sequence: stats -> bigger image -> stats -> crops of the new image must still be the image's bytes."""
    g = torch.Generator().manual_seed(11)
    low = (torch.randn(192, 192, generator=g) * 3).numpy()
    rects = [[0, 0, 100, 100], [50, 60, 70, 80]]
    vsm.set_image(synthetic_image(320, 240, 1))
    first = vsm.heatmap_stats(low, 240, 320, rects)
    for k, (w, h) in enumerate([(1600, 1200), (3840, 2160)]):           # two successive regrows
        img = synthetic_image(w, h, 40 + k)
        vsm.set_image(img)
        again = vsm.heatmap_stats(low, 240, 320, rects)
        assert np.array_equal(np.asarray(first), np.asarray(again))
        box = [w - 300, h - 260, w, h]
        clip_gpu, _ = vsm.engine.preprocess_only([box])
        ref = torch.from_numpy(pp.clip_preprocess(img.crop(tuple(box)), vsm.cfg.clip_image_size)).bfloat16().float().numpy()
        assert np.array_equal(clip_gpu[0], ref)                           # the resident image was not overwritten


def test_stream_with_default_vsm_settings_batches_across_targets(vsm):
    """the SHIPPED default (group_prompts = True; visual_search.py / vstar_bench_eval.py never change it) under
    visual_search_stream: searches for different objects on different images share engine calls (one call per scoring round while
    the round fits the batch), and every result is bit-identical to the per-sample loop run through the same (grouped) entry
    point."""
    from vstar_amd.search import visual_search_stream
    sizes = [(1280, 720), (900, 1100), (1500, 640)]
    imgs = [synthetic_image(w, h, 80 + k) for k, (w, h) in enumerate(sizes)]
    names = ["kite", "dog", "small red umbrella", "boat", "traffic light", "cup"]
    samples = [(imgs[k % 3], n, None, smallest_size_for(*sizes[k % 3])) for k, n in enumerate(names)]
    kw = dict(confidence_high=2.0, confidence_low=0.0, target_cue_threshold=-1.0, target_cue_threshold_minimum=-1.0)
    assert vsm.group_prompts is True
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        st = {}
        got = visual_search_stream(vsm, samples, window=6, stats=st, **kw)
        vsm.group_prompts = "always"
        try:
            loop = [visual_search(vsm, im, n, gt, sm, speculate=False, **kw) for im, n, gt, sm in samples]
        finally:
            vsm.group_prompts = True
    for x, y in zip(loop, got):
        assert x[1] == y[1] and x[2] == y[2] and x[0]["bbox"] == y[0]["bbox"]
        assert torch.equal(x[0]["detection_result"], y[0]["detection_result"])
    # no per-prompt fragmentation: a step is one call unless its crops exceed the grouped entry point's activation-row budget
    # (6 one-prompt crops per call in this tiny configuration); round 3 made one call per distinct prompt = one per crop here
    # (a step that holds more than 6 crops — its live searches plus the policy's speculation — is two calls)
    assert st["engine_steps"] <= st["engine_calls"] <= 2 * st["engine_steps"] and st["engine_calls"] < st["crops_scored"] / 2
    assert st["crops_scored"] / st["engine_calls"] > 2.0          # (one call per prompt would be exactly 1.0)


def test_search_with_device_reductions_equals_host_path(vsm):
    img = synthetic_image(1280, 720, 33)
    smallest = smallest_size_for(1280, 720)
    kw = dict(confidence_high=2.0, confidence_low=0.0, target_cue_threshold=-1.0, target_cue_threshold_minimum=-1.0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a, b = {}, {}
        r_host = visual_search(vsm, img, "kite", None, smallest, stats=a, **kw)
        r_dev = visual_search(vsm, img, "kite", None, smallest, stats=b, device_reductions=True, **kw)
    assert r_host[1] == r_dev[1] and r_host[0]["bbox"] == r_dev[0]["bbox"]
    assert torch.equal(r_host[0]["detection_result"], r_dev[0]["detection_result"])
    assert [p["bbox"] for p in a["search_path"]] == [p["bbox"] for p in b["search_path"]]
    sa = [float(p["score"]) for p in a["search_path"][1:]]
    sb = [float(p["score"]) for p in b["search_path"][1:]]
    assert np.allclose(sa, sb, rtol=1e-5, atol=1e-7)


def test_cross_image_stream_equals_per_sample_loop_and_mixes_images(vsm):
    """visual_search_stream on the engine: (image, target) samples of THREE different images searched in a window
    — crops of different images in the same engine batch through the image slots — give bit-identical results to the reference's
    one-sample-at-a-time loop; the GPU preprocessing reads every crop from its own slot."""
    from vstar_amd.search import visual_search_stream
    sizes = [(1280, 720), (900, 1100), (1500, 640)]
    imgs = [synthetic_image(w, h, 70 + k) for k, (w, h) in enumerate(sizes)]
    samples = [(imgs[0], "kite", None, smallest_size_for(*sizes[0])), (imgs[1], "dog", None, smallest_size_for(*sizes[1])),
               (imgs[0], "small red umbrella", None, smallest_size_for(*sizes[0])), (imgs[2], "boat", None, smallest_size_for(*sizes[2])),
               (imgs[1], "traffic light", None, smallest_size_for(*sizes[1]))]
    kw = dict(confidence_high=2.0, confidence_low=0.0, target_cue_threshold=-1.0, target_cue_threshold_minimum=-1.0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        vsm.group_prompts = False
        try:
            loop = [visual_search(vsm, im, n, gt, sm, speculate=False, **kw) for im, n, gt, sm in samples]
            st = {}
            got = visual_search_stream(vsm, samples, window=4, stats=st, **kw)
            # slot-addressed preprocessing == slot 0 preprocessing of the same image
            vsm.set_image(imgs[1], 5)
            vsm.set_image(imgs[1], 0)
            a = vsm.engine.preprocess_only(np.asarray([[10, 20, 500, 700]], np.int32), [5])
            b = vsm.engine.preprocess_only(np.asarray([[10, 20, 500, 700]], np.int32))
        finally:
            vsm.group_prompts = True
    for x, y in zip(loop, got):
        assert x[1] == y[1] and x[2] == y[2] and x[0]["bbox"] == y[0]["bbox"]
        assert torch.equal(x[0]["detection_result"], y[0]["detection_result"])
    assert st["searches"] == 5
    assert st["crops_scored"] >= st["useful_crops"] > 5 and 0.0 <= st["wasted_crop_frac"] < 1.0
    assert st["engine_steps"] < st["useful_crops"]            # several searches per engine step
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
