"""Full-depth, real-width parity at the headline-bench shape.

tests/golden/full7b_336.npz holds the REFERENCE's own model_forward(inference=True) outputs (fp32, and bf16 as the noise
yardstick) for EIGHT crops of the exact 32-crop batch bench.py scores, at CLIP-L/14@336 x 23 blocks, LLaMA-7B x 32 layers (S=640),
OWL-ViT-B/16@768 x 12 layers + SAM head, with the bench's weights (oracle/gen_fulldepth_golden.py); full7b_224.npz the same for
four crops at the geometry the reference REALLY runs (CLIP-L/14@224, 256 image tokens, S = 320: VisualSearch/model/VSM.py:230-234,
466-473).  Here the engine scores the WHOLE batch (B=32, the tested state dict IS the bench's) and the recorded crops must land
within 1.5 x the reference's own bf16 noise per tap, with the same arg-max box / margin-aware top-k order and teacher-forced
arg-max tokens; each recorded crop scored ALONE (B = 1) is bit-identical to its in-batch record; the scheduler's decisions
(tests/_parity.py::decisions) agree with the fp32 reference as often as the reference's own bf16 run does; and the UN-centred
mask error pooled over the recorded crops stays within 1.5 x the reference-bf16's."""
import os

import numpy as np
import pytest
import torch

from _parity import assert_mask_within_bf16_noise, assert_within_bf16_noise, decision_agreement, decisions, fmt, rel_l2
from test_engine_gpu import margin_aware_topk_equal
from vstar_amd.config import VSMConfig
from vstar_amd.engine import VstarEngine
from vstar_amd.synthetic import bench_inputs
from vstar_amd.preprocess import SyntheticTokenizer
from vstar_amd.weights import random_state_dict, template_chain, trained_like_state_dict

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _assert(cond, msg):
    assert cond, msg


def _state_dict(cfg, z):
    """The state dict a golden file was recorded with: i.i.d. random (rounds 2/3) or the trained-like statistics of round 4 —
    outlier residual channels (|x|_inf / rms 9 -> 32 across the depth, 48 at the worst token), a massive-activation BOS, spread
    norm gains, peaked attention (vstar_amd.weights.trained_like_state_dict; oracle/gen_fulldepth_golden.py --weights trained_like)."""
    if "weights" in z.files and str(z["weights"]) == "trained_like":
        return trained_like_state_dict(cfg, seed=int(z["weight_seed"]), dtype=torch.bfloat16, share_layers=True,
                                       chain=template_chain(SyntheticTokenizer(cfg.llm_vocab)))
    return random_state_dict(cfg, seed=int(z["weight_seed"]), dtype=torch.bfloat16, share_layers=True)


@pytest.mark.parametrize("image_size,weights,fold", [(336, "random", None), (224, "random", None),
                                                     (336, "trained_like", "1"), (336, "trained_like", "0"),
                                                     (224, "trained_like", "1"), (224, "trained_like", "0")])
def test_bench_batch_matches_full_depth_reference(cuda, image_size, weights, fold, monkeypatch):
    """weights = trained_like (VERDICT r3 missing #2): the same gates on weights with the statistics of a trained checkpoint, pushed
    through the reference's own model_forward in fp32 and bf16, for BOTH norm formulations of the engine (VSTAR_FOLD_NORMS = 1: rstd
    applied to the fp32 accumulators of the raw residual stream — the variant whose headroom outlier channels could eat — and = 0:
    the reference's rounding points)."""
    tag = "tl_" if weights == "trained_like" else ""
    z = np.load(os.path.join(GOLD, f"full7b_{tag}{image_size}.npz"))
    B, T = int(z["batch"]), int(z["text_tokens"])
    assert len(z["crops"]) >= (4 if image_size == 224 or tag else 8)
    if fold is not None:
        monkeypatch.setenv("VSTAR_FOLD_NORMS", fold)
    cfg = VSMConfig.seal_7b(image_size, max_batch=B, max_text_len=T + 1)
    eng = VstarEngine(cfg, 0)
    eng.load_state_dict(_state_dict(cfg, z))
    clip, owl, ids, loc, verify = bench_inputs(cfg, B, T)
    out = eng.score_batch(clip.to(cuda), owl.to(cuda), ids, loc, verify_pos=verify)
    H = cfg.llm_hidden
    hidden = eng.debug_read("llm_hidden_loc", B * H).reshape(B, H)
    det = eng.debug_read("embed_det", B * 512).reshape(B, 512)
    seg = eng.debug_read("embed_seg", B * 256).reshape(B, 256)
    hyper = eng.debug_read("sam_hyper", B * 32).reshape(B, 32)
    dec = {"engine": [], "fp32": [], "bf16": []}
    fails = []                                # every tap of every crop is evaluated before the verdict: one run shows the whole picture

    def gate(fn, *a, **k):
        try:
            fn(*a, **k)
        except AssertionError as exc:
            fails.append(str(exc).splitlines()[0])

    # Per-crop errors are a heavy-tailed statistic on the OWL-ViT side: OwlViT multiplies every patch token by the class token
    # (owlvit.py:128-138), so the rounding luck of ONE row scales the error of a whole crop — the reference's own bf16 run varies
    # by 2 x from crop to crop on the same tap (sam_upscaled_mean 6.3e-3 ... 1.2e-2 on the trained-like fixture) and the torch-bf16
    # oracle shows the same spread over the 32 crops of the batch (tools/owl_crop_spread.py -> profiles/r04_owl_crop_spread.json).
    # So a tap is gated (a) POOLED over the fixture's crops at 1.5 x the pooled reference-bf16 noise — the rule of tests/_parity.py —
    # and (b) per crop at 1.5 x the LARGEST noise the reference's own bf16 run shows on any crop of the fixture.
    per_tap = {}                              # tap -> [(engine error, reference-bf16 noise)] per crop
    off_sig = []                              # (engine, reference-bf16) mask offsets in units of the noise model's sigma
    for j, ci in enumerate(z["crops"]):
        ci = int(ci)
        rep = {}
        got = {"llm_hidden_loc": hidden[ci], "embed_det": det[ci], "embed_seg": seg[ci], "sam_hyper": hyper[ci],
               "pred_logits": out["pred_logits"][ci, :, 0], "pred_boxes": out["pred_boxes"][ci]}
        got["sam_upscaled_mean"] = eng.debug_read("sam_c2", (ci + 1) * 192 * 192 * 32)[ci * 192 * 192 * 32:].reshape(-1, 32).astype(
            np.float64).mean(axis=0)
        for k, v in got.items():
            assert np.isfinite(v).all(), k
            rep[k] = (rel_l2(v, z[k][j]), rel_l2(z["bf16_" + k][j], z[k][j]))
            per_tap.setdefault(k, []).append(rep[k])
        mrep = {}
        try:
            assert_mask_within_bf16_noise(out["low_res_masks"][ci, 0], z["low_res_masks"][j], z["bf16_low_res_masks"][j],
                                          z["sam_hyper"][j], z["bf16_sam_hyper"][j], z["sam_upscaled_mean"][j],
                                          z["bf16_sam_upscaled_mean"][j], factor=1e9, report=mrep)
        except AssertionError:
            pass                              # (its 3-sigma offset rule is re-stated below on the pooled numbers)
        per_tap.setdefault("mask_pattern", []).append(mrep["mask_pattern"])
        off_sig.append(mrep["mask_offset_sigma[0]"])
        rep.update(mrep)
        print(f"\ncrop {ci}: engine / reference-bf16 noise (rel-L2 vs the reference's fp32 output): {fmt(rep)}")
        # largest single box error: 1e-2, or 1.5 x the largest the reference's own bf16 run shows on this fixture's crops (trained-like
        # weights: up to 1.08e-2 for the reference itself)
        box_cap = max(1e-2, 1.5 * float(np.abs(z["bf16_pred_boxes"] - z["pred_boxes"]).max()))
        gate(lambda: _assert(np.abs(out["pred_boxes"][ci] - z["pred_boxes"][j]).max() < box_cap, f"crop {ci}: max box error over {box_cap:.2e}"))
        noise_abs = 2.0 * float(np.abs(z["bf16_pred_logits"][j] - z["pred_logits"][j]).max())
        ok, msg = margin_aware_topk_equal(got["pred_logits"], z["pred_logits"][j], 5, noise_abs)
        assert ok, msg
        # teacher-forced arg-max at the three verify positions: equal to the reference's unless its own top-2 gap is inside the
        # logit noise (the reference's bf16 run is the judge of that: where IT flips, the engine may)
        for v in range(verify.shape[1]):
            if int(out["tf_argmax"][ci, v]) != int(z["tf_argmax"][j, v]):
                assert float(z["tf_top2_gap"][j, v]) <= 2e-2 * float(z["tf_logit_spread"][j, v]), (ci, v)
        dec["engine"].append(decisions(out["pred_logits"][ci, :, 0], out["pred_boxes"][ci], out["low_res_masks"][ci, 0]))
        dec["fp32"].append(decisions(z["pred_logits"][j], z["pred_boxes"][j], z["low_res_masks"][j]))
        dec["bf16"].append(decisions(z["bf16_pred_logits"][j], z["bf16_pred_boxes"][j], z["bf16_low_res_masks"][j]))
    rms = lambda xs: float(np.sqrt(np.mean(np.square(xs))))  # noqa: E731
    for k, pairs in per_tap.items():
        e, n = np.asarray([p[0] for p in pairs]), np.asarray([p[1] for p in pairs])
        print(f"{k:18s} pooled engine {rms(e):.3e} / reference-bf16 {rms(n):.3e} (x{rms(e) / rms(n):.2f}); worst crop x{(e / n).max():.2f}, "
              f"engine max {e.max():.3e} vs reference-bf16 max {n.max():.3e}")
        # (round 5: 1.35, was 1.5.  Over ALL 32 crops the ratio is 0.93 - 1.02 on every tap and gated at 1.25 — the x32 test below;
        # a ratio of two rms values over only 4 - 8 heavy-tailed crops scatters by +-20 % from sampling alone, hence not 1.25 here)
        gate(lambda: _assert(rms(e) <= 1.35 * rms(n), f"{k}: pooled engine {rms(e):.2e} vs reference-bf16 {rms(n):.2e} (x{rms(e) / rms(n):.2f} > 1.35)"))
        gate(lambda: _assert(e.max() <= 1.5 * n.max(), f"{k}: worst crop {e.max():.2e} vs the reference-bf16's worst {n.max():.2e}"))
    # mask offset in units of the noise model's sigma (tests/_parity.py): a unit Gaussian would give rms 1 and rarely exceed 3; the
    # reference's own bf16 run is the yardstick for the pooled value, 4 sigma the per-crop bound
    eo, no = np.asarray([p[0] for p in off_sig]), np.asarray([p[1] for p in off_sig])
    print(f"mask offset / sigma: engine rms {rms(eo):.2f} max {eo.max():.2f}; reference-bf16 rms {rms(no):.2f} max {no.max():.2f}")
    gate(lambda: _assert(rms(eo) <= max(1.5, 1.75 * rms(no)) and eo.max() <= 4.0, f"mask offsets: engine rms {rms(eo):.2f} max {eo.max():.2f} sigma"))
    assert not fails, "\n".join(fails)
    # ---- pooled over the recorded crops: un-centred mask error and the scheduler's decisions, next to the reference's bf16 run ----
    sel = [int(c) for c in z["crops"]]
    e_mask, n_mask = rel_l2(out["low_res_masks"][sel, 0], z["low_res_masks"]), rel_l2(z["bf16_low_res_masks"], z["low_res_masks"])
    print(f"\npooled un-centred mask rel-L2 over {len(sel)} crops: engine {e_mask:.3e} / reference-bf16 {n_mask:.3e}")
    assert e_mask <= 1.5 * n_mask, (e_mask, n_mask)
    rep_e, rep_b = decision_agreement(dec["engine"], dec["fp32"]), decision_agreement(dec["bf16"], dec["fp32"])
    print("decisions, engine vs reference-fp32:        ", {k: round(v, 4) for k, v in rep_e.items()})
    print("decisions, reference-bf16 vs reference-fp32:", {k: round(v, 4) for k, v in rep_b.items()})
    slack = 1.0 / len(sel)                    # one crop: the granularity of these rates
    for k, v in rep_e.items():
        if k.endswith("_same") or k.startswith("argmax_box"):
            assert v >= rep_b[k] - slack - 1e-9, f"{k}: engine {v:.3f} vs reference-bf16 {rep_b[k]:.3f}"
    # (pos_frac counts pixels of a 192 x 192 map: one pixel = 2.7e-5 is its granularity — with nearly-all-negative masks both runs
    # differ from fp32 by about one pixel)
    # The continuous companions of those decisions are gated at 1.5 x the reference-bf16's value PLUS a magnitude that cannot move a
    # decision (the decisions themselves — orders, arg-max, threshold crossings — are gated above): 1e-3 of a child's share of the
    # heat mass, 1 % of the heat map's maximum.  On the trained-like fixtures the masks are 99.5 % negative, so these statistics
    # rest on a few hundred pixels of four crops and the reference's own values are tiny (child shares 2e-4, score max 0.8 %).
    for k, slack_abs in (("child_share_rms_diff", 1e-3), ("pos_frac_rms_diff", 1.0 / (192 * 192)), ("score_max_rel_rms", 1e-2)):
        assert rep_e[k] <= 1.5 * rep_b[k] + slack_abs, f"{k}: engine {rep_e[k]:.3e} vs reference-bf16 {rep_b[k]:.3e}"
    # batch invariance at the bench shape: every recorded crop scored ALONE (B = 1, the latency regime of a sharded search: other
    # GEMM kernels / tile shapes than at B = 32) is bit-identical to its record inside the 32-crop batch
    for ci in sel[:3]:
        solo = eng.score_batch(clip[ci:ci + 1].to(cuda), owl[ci:ci + 1].to(cuda), ids[ci:ci + 1], loc[ci:ci + 1], verify_pos=verify[ci:ci + 1])
        for k in ("pred_logits", "pred_boxes", "low_res_masks", "tf_argmax"):
            assert np.array_equal(solo[k][0], out[k][ci]), (ci, k)
    eng.close()


@pytest.mark.parametrize("image_size,tag", [(336, "x32"), (224, "x32"), (336, "x32_r1")])
def test_all_32_crops_engine_noise_equals_reference_bf16_noise(cuda, image_size, tag):
    """Round 5: the 8-crop fixtures above cannot tell a 10 % systematic excess from sampling error — the
    per-crop errors on the OWL-ViT side are heavy-tailed and the pooled engine / reference-bf16 ratio moved between 0.72 and 1.21
    from fixture to fixture.  tests/golden/full7b_tl_336_x32.npz holds the reference's fp32 AND bf16 outputs for ALL 32 crops of the
    bench batch (trained-like weights; oracle/gen_fulldepth_golden.py --weights trained_like --crops all --mask-f16).  Gates, pooled
    (rms) over the 32 crops:
      * every tap: engine error vs fp32 <= 1.25 x the reference-bf16's own error (measured 0.93 - 1.02; round 4 gated 1.5);
      * the DIRECT distance engine <-> reference-bf16 <= 1.25 x sqrt(2) x that noise: two independent roundings of one fp32 result
        sit sqrt(2) noise units apart, a systematic difference between the two bf16 evaluations would show as more;
      * mask offset (units of the noise model's sigma with ONE noise level per fixture, see below): engine rms <= 1.25 x the
        reference-bf16's + 0.1, no crop beyond max(4, 1.25 x the reference's worst);
      * the signed mask offsets average to zero within 3 standard errors (no bias of the engine against fp32)."""
    path = os.path.join(GOLD, f"full7b_tl_{image_size}_{tag}.npz")           # 224: the geometry the reference REALLY runs (S = 320)
    if not os.path.exists(path):
        pytest.skip(f"{os.path.basename(path)} not generated")
    z = np.load(path)                                                          # x32_r1 (round 6): a second crop batch, bench_inputs(rank=1)
    B, T = int(z["batch"]), int(z["text_tokens"])
    crops = [int(c) for c in z["crops"]]
    assert len(crops) == B == 32
    cfg = VSMConfig.seal_7b(image_size, max_batch=B, max_text_len=T + 1)
    eng = VstarEngine(cfg, 0)
    eng.load_state_dict(_state_dict(cfg, z))
    clip, owl, ids, loc, verify = bench_inputs(cfg, B, T, rank=int(z["input_rank"]) if "input_rank" in z.files else 0)
    out = eng.score_batch(clip.to(cuda), owl.to(cuda), ids, loc, verify_pos=verify)
    H = cfg.llm_hidden
    taps = {"llm_hidden_loc": eng.debug_read("llm_hidden_loc", B * H).reshape(B, H),
            "embed_det": eng.debug_read("embed_det", B * 512).reshape(B, 512),
            "embed_seg": eng.debug_read("embed_seg", B * 256).reshape(B, 256),
            "sam_hyper": eng.debug_read("sam_hyper", B * 32).reshape(B, 32),
            "pred_logits": out["pred_logits"][:, :, 0], "pred_boxes": out["pred_boxes"],
            "sam_upscaled_mean": eng.debug_read("sam_c2", B * 192 * 192 * 32).reshape(B, -1, 32).astype(np.float64).mean(axis=1)}
    rms = lambda xs: float(np.sqrt(np.mean(np.square(xs))))  # noqa: E731
    fails = []
    for k, got in taps.items():
        e = rms([rel_l2(got[ci], z[k][j]) for j, ci in enumerate(crops)])
        n = rms([rel_l2(z["bf16_" + k][j], z[k][j]) for j in range(B)])
        d = rms([rel_l2(got[ci], z["bf16_" + k][j]) for j, ci in enumerate(crops)])
        print(f"{k:18s} engine {e:.3e} / reference-bf16 {n:.3e} (x{e / n:.3f}); engine <-> reference-bf16 {d:.3e} = {d / (np.sqrt(2) * n):.3f} x sqrt(2) noise")
        if not e <= 1.25 * n:
            fails.append(f"{k}: pooled engine {e:.3e} vs reference-bf16 {n:.3e} (x{e / n:.2f} > 1.25)")
        if not d <= 1.25 * np.sqrt(2.0) * n:
            fails.append(f"{k}: engine <-> reference-bf16 {d:.3e} vs sqrt(2) x noise {np.sqrt(2) * n:.3e}")
    m32, m16 = z["low_res_masks"].astype(np.float64), z["bf16_low_res_masks"].astype(np.float64)
    mg = out["low_res_masks"][crops, 0].astype(np.float64)
    pat = lambda m: m - m.mean(axis=(1, 2), keepdims=True)  # noqa: E731
    e = rms([rel_l2(pat(mg[j:j + 1]), pat(m32[j:j + 1])) for j in range(B)])
    n = rms([rel_l2(pat(m16[j:j + 1]), pat(m32[j:j + 1])) for j in range(B)])
    print(f"mask_pattern       engine {e:.3e} / reference-bf16 {n:.3e} (x{e / n:.3f})")
    if not e <= 1.25 * n:
        fails.append(f"mask pattern x{e / n:.2f}")
    eu, nu = rel_l2(mg, m32), rel_l2(m16, m32)
    print(f"mask un-centred    engine {eu:.3e} / reference-bf16 {nu:.3e} (x{eu / nu:.3f})")
    if not eu <= 1.25 * nu:
        fails.append(f"un-centred mask x{eu / nu:.2f}")
    ze, zn, signed = [], [], []
    for j in range(B):
        r = {}
        try:
            assert_mask_within_bf16_noise(mg[j], m32[j], m16[j], z["sam_hyper"][j], z["bf16_sam_hyper"][j], z["sam_upscaled_mean"][j],
                                          z["bf16_sam_upscaled_mean"][j], factor=1e9, report=r)
        except AssertionError:
            pass
        ze.append(r["mask_offset_sigma[0]"][0])
        zn.append(r["mask_offset_sigma[0]"][1])
        signed.append(float(mg[j].mean() - m32[j].mean()))
    print(f"mask offset / sigma (per-crop unit): engine rms {rms(ze):.2f} max {max(ze):.2f}; reference-bf16 rms {rms(zn):.2f} max {max(zn):.2f}; "
          f"signed engine offsets mean {np.mean(signed):+.4f} +- {np.std(signed) / np.sqrt(B):.4f}")
    # Round 6: the unit of that comparison was unfair to the engine.  sigma of crop j is scaled by the reference-bf16 run's OWN relative
    # error on crop j (tests/_parity.py), which varies 2x from crop to crop (0.009 - 0.021): for the reference's offsets numerator and
    # denominator come from the same run and are correlated, for the engine the denominator is an independent random scale — a
    # ratio distribution with heavy tails (second fixture: one crop at 5.08 "sigma" whose unit was the fixture's smallest).  The gate
    # therefore uses ONE noise level per fixture, the rms of the per-crop levels, for both sides: the reference's own offsets then
    # read rms 1.26 / max 3.82 (first fixture) and 1.01 / 3.04 (second).
    eps = np.asarray([np.hypot(rel_l2(z["bf16_sam_hyper"][j], z["sam_hyper"][j]), rel_l2(z["bf16_sam_upscaled_mean"][j], z["sam_upscaled_mean"][j]))
                      for j in range(B)])
    unit = np.asarray([np.sqrt(np.mean(eps ** 2)) * np.linalg.norm(z["sam_hyper"][j].astype(np.float64)) *
                       np.linalg.norm(z["sam_upscaled_mean"][j].astype(np.float64)) / np.sqrt(z["sam_hyper"].shape[-1]) for j in range(B)])
    pe = np.abs(np.asarray(signed)) / unit
    pn = np.abs(m16.mean(axis=(1, 2)) - m32.mean(axis=(1, 2))) / unit
    print(f"mask offset / sigma (fixture unit):  engine rms {rms(pe):.2f} max {pe.max():.2f}; reference-bf16 rms {rms(pn):.2f} max {pn.max():.2f}")
    _POOLED_Z[(image_size, tag)] = (pe, pn)
    if not (rms(pe) <= 1.25 * rms(pn) + 0.1 and pe.max() <= max(4.0, 1.25 * pn.max())):
        fails.append(f"mask offsets: engine rms {rms(pe):.2f} max {pe.max():.2f} vs reference-bf16 rms {rms(pn):.2f} max {pn.max():.2f} (fixture unit)")
    if not abs(np.mean(signed)) <= 3.0 * np.std(signed) / np.sqrt(B):
        fails.append(f"mask offsets are biased: mean {np.mean(signed):+.4f}")
    eng.close()
    _SIGNED_OFFSETS[(image_size, tag)] = signed
    assert not fails, "\n".join(fails)


_SIGNED_OFFSETS = {}
_POOLED_Z = {}


def test_mask_offset_is_unbiased_over_both_32_crop_batches(cuda):
    """Round 6: over the 32 crops of the first 336 fixture the engine's signed mask offset against the reference's
    fp32 masks was -0.0088 +- 0.0044 (two standard errors).  A second 32-crop batch on another input seed
    (tests/golden/full7b_tl_336_x32_r1.npz) settles it: pooled over the 64 crops |mean| <= 2 standard errors.  Uses the offsets
    the parametrised test above recorded in this session (same engine runs)."""
    a, b = _SIGNED_OFFSETS.get((336, "x32")), _SIGNED_OFFSETS.get((336, "x32_r1"))
    if a is None or b is None:
        pytest.skip("needs both 336 fixtures to have run in this session")
    s = np.asarray(a + b, np.float64)
    se = s.std() / np.sqrt(len(s))
    print(f"signed mask offsets over {len(s)} crops: mean {s.mean():+.4f} +- {se:.4f}  (batch 0: {np.mean(a):+.4f}, batch 1: {np.mean(b):+.4f})")
    assert abs(s.mean()) <= 2.0 * se, f"mask offsets are biased over 64 crops: mean {s.mean():+.4f}, standard error {se:.4f}"
    pe = np.concatenate([_POOLED_Z[(336, "x32")][0], _POOLED_Z[(336, "x32_r1")][0]])
    pn = np.concatenate([_POOLED_Z[(336, "x32")][1], _POOLED_Z[(336, "x32_r1")][1]])
    r = lambda x: float(np.sqrt(np.mean(np.square(x))))  # noqa: E731
    print(f"mask offset / sigma over 64 crops (fixture units): engine rms {r(pe):.2f} max {pe.max():.2f}; reference-bf16 rms {r(pn):.2f} max {pn.max():.2f}")
    assert r(pe) <= 1.25 * r(pn), "the engine's mask offsets are noisier than the reference's own bf16 run over 64 crops"
