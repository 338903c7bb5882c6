"""Shared parity gate: a bf16 engine is judged against the bf16 noise of the SAME algorithm, measured, not
against a fixed floor.  `ref32` is the fp32 evaluation (reference golden or oracle), `ref16` the reference / oracle evaluated in
bf16 on torch-CPU; the engine must land within `factor` x the bf16 evaluation's own distance from fp32."""
import numpy as np




def rel_l2(got, ref):
    got = np.asarray(got, dtype=np.float64).reshape(-1)
    ref = np.asarray(ref, dtype=np.float64).reshape(-1)
    return float(np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-30))


def assert_within_bf16_noise(name, got, ref32, ref16, factor=1.5, report=None):
    e, n = rel_l2(got, ref32), rel_l2(ref16, ref32)
    if report is not None:
        report[name] = (e, n)
    assert np.isfinite(e) and e <= factor * n, f"{name}: engine {e:.2e} vs bf16 noise {n:.2e} (x{e / max(n, 1e-30):.2f} > {factor})"
    return e, n


def assert_mask_within_bf16_noise(got, ref32, ref16, hyper32, hyper16, upmean32, upmean16, factor=1.5, report=None):
    """Masks [B,192,192] (or one mask): pattern (per-mask mean removed) at the `factor` rule pooled over the batch; offset
    inside the band of the noise model sigma = sqrt(eps_h^2 + eps_mu^2) |h| |mu| / sqrt(32) per mask (4 sigma each, rms pooled; see
    tests/test_engine_gpu.py::test_engine_matches_reference_golden).  hyper*/upmean*: [B,32] operands of the final product."""
    f = lambda a, nd: np.asarray(a, np.float64).reshape((-1,) + tuple(np.asarray(a).shape[-nd:]))  # noqa: E731
    got, ref32, ref16 = f(got, 2), f(ref32, 2), f(ref16, 2)
    h32, h16, m32, m16 = f(hyper32, 1), f(hyper16, 1), f(upmean32, 1), f(upmean16, 1)
    pat = lambda m: m - m.mean(axis=(1, 2), keepdims=True)  # noqa: E731
    e, n = rel_l2(pat(got), pat(ref32)), rel_l2(pat(ref16), pat(ref32))
    if report is not None:
        report["mask_pattern"] = (e, n)
    assert e <= factor * n, f"mask pattern: engine {e:.2e} vs bf16 noise {n:.2e}"
    # Offsets in units of the noise model's sigma.  Round 4: the suite gates ~100 masks; a per-mask 3-sigma bound alone raises a
    # false alarm in one run of four even for a perfect implementation (0.27 % per mask under an ideal Gaussian, more with real
    # tails) — a change of rounding points (the folded ViT LayerNorms) moved one tiny-model mask from 2.9 to 3.08 sigma.  The bound is
    # therefore 4 sigma per mask (6e-5 each) AND the rms over the call's n masks within 1 + 3 / sqrt(2 n) of its expectation 1
    # (three standard deviations of the rms of n unit Gaussians: 3.1 for one mask, 2.06 for four, 1.375 for thirty-two).
    zs, zs16 = [], []
    for b in range(got.shape[0]):
        eps = np.hypot(rel_l2(h16[b], h32[b]), rel_l2(m16[b], m32[b]))
        sigma = eps * np.linalg.norm(h32[b]) * np.linalg.norm(m32[b]) / np.sqrt(h32.shape[-1])
        off = abs(got[b].mean() - ref32[b].mean())
        off16 = abs(ref16[b].mean() - ref32[b].mean())
        if report is not None:
            report[f"mask_offset_sigma[{b}]"] = (off / sigma, off16 / sigma)
        zs.append(off / sigma)
        zs16.append(off16 / sigma)
        assert off <= 4.0 * sigma, f"mask {b}: offset {off:.4f} outside 4 sigma = {4 * sigma:.4f} of the bf16 noise model"
    # The UNIT is calibrated PER FIXTURE (round 6; the round-5 global constant 1.2 loosened every caller's gate).  The noise model's
    # sigma is a first-order estimate: measured against it, the reference's OWN bf16 run on the same masks has offset rms `rms16`
    # (1.12 over the 32 crops of full7b_tl_336_x32.npz, 1.28 - 1.32 over the 8-crop fixtures) where a perfectly scaled sigma would
    # give 1.00.  The engine is therefore bounded by THAT run: rms <= max(1, rms16) * (1 + 3 / sqrt(2 n)) — three standard deviations
    # of the rms of n Gaussians of the scale the reference's bf16 evaluation shows on these very masks, never tighter than the ideal
    # model; the per-mask 4-sigma cap above is unchanged.
    n = len(zs)
    rms = float(np.sqrt(np.mean(np.square(zs))))
    rms16 = float(np.sqrt(np.mean(np.square(zs16))))
    if report is not None:
        report["mask_offset_rms"] = (rms, rms16)
    bound = max(1.0, rms16) * (1.0 + 3.0 / np.sqrt(2.0 * n))
    assert rms <= bound, f"mask offsets: rms {rms:.2f} sigma over {n} masks (bound {bound:.2f}; reference bf16 {rms16:.2f})"


def fmt(report):
    return "  ".join(f"{k} {a:.2e}/{b:.2e}" for k, (a, b) in report.items())


# ---------------- decision-level parity (what the scheduler actually reads) ----------------
# The search loop (visual_search.py:390-516) consumes a crop's outputs only through a handful of decisions:
#   * detection: arg-max box of the bf16 sigmoid scores, `top > confidence_high (0.5)`, `top >= confidence_low (0.3)`;
#   * heat map:  clamp(mask, 0).max() > threshold (6.0 * 0.7^(level-1), floor 3.0) -> cue branch or not;
#                min-max normalised mass of the four child rectangles -> child ORDER in the priority queue.
# `decisions()` evaluates them on one crop's raw outputs (on the 192^2 low-res map: the bilinear upsample is a fixed linear map
# applied to both sides), `decision_agreement()` compares two evaluations of the same crops.
CUE_THRESHOLDS = (6.0, 6.0 * 0.7, 3.0)


def decisions(pred_logits, pred_boxes, low_res_mask):
    import torch
    lg = np.asarray(pred_logits, np.float32).reshape(-1)
    scores = torch.from_numpy(lg).to(torch.bfloat16).sigmoid().float().numpy()       # visual_search.py:225 keeps bf16
    top = int(scores.argmax())                                                          # first maximum, like tensor.argmax()
    heat = np.maximum(np.asarray(low_res_mask, np.float64).reshape(192, 192), 0.0)      # visual_search.py:224 clamp(min=0)
    mx, mn = float(heat.max()), float(heat.min())
    norm = (heat - mn) / (mx - mn) if mx != mn else heat * 0
    tot = norm.sum()
    quads = np.asarray([norm[:96, :96].sum(), norm[:96, 96:].sum(), norm[96:, :96].sum(), norm[96:, 96:].sum()])
    shares = quads / tot if tot > 0 else quads * 0
    return {"top_index": top, "top_score": float(scores[top]), "top_box": np.asarray(pred_boxes, np.float64).reshape(-1, 4)[top],
            "n_valid": int((scores > 0.5).sum()), "score_max": mx, "pos_frac": float((heat > 0).mean()), "child_shares": shares,
            "child_order": tuple(np.argsort(-shares, kind="stable"))}


def decision_agreement(a, b, score_thresholds=(0.5, 0.3), cue_thresholds=CUE_THRESHOLDS):
    """Identity rates of the scheduler's decisions between two evaluations `a`, `b` (lists of decisions()) of the same crops."""
    n = len(a)
    box_iou = []
    for x, y in zip(a, b):
        bx, by = x["top_box"], y["top_box"]
        x1, y1 = max(bx[0] - bx[2] / 2, by[0] - by[2] / 2), max(bx[1] - bx[3] / 2, by[1] - by[3] / 2)
        x2, y2 = min(bx[0] + bx[2] / 2, by[0] + by[2] / 2), min(bx[1] + bx[3] / 2, by[1] + by[3] / 2)
        inter = max(0.0, x2 - x1) * max(0.0, y2 - y1)
        box_iou.append(inter / max(bx[2] * bx[3] + by[2] * by[3] - inter, 1e-12))
    rep = {"n": n,
           "argmax_box_same_index": float(np.mean([x["top_index"] == y["top_index"] for x, y in zip(a, b)])),
           "argmax_box_iou_ge_0.9": float(np.mean([v >= 0.9 for v in box_iou])),
           "child_order_same": float(np.mean([x["child_order"] == y["child_order"] for x, y in zip(a, b)])),
           "best_child_same": float(np.mean([x["child_order"][0] == y["child_order"][0] for x, y in zip(a, b)])),
           "child_share_max_abs_diff": float(np.max([np.abs(x["child_shares"] - y["child_shares"]).max() for x, y in zip(a, b)])),
           "child_share_rms_diff": float(np.sqrt(np.mean([((x["child_shares"] - y["child_shares"]) ** 2).mean() for x, y in zip(a, b)]))),
           "pos_frac_max_abs_diff": float(np.max([abs(x["pos_frac"] - y["pos_frac"]) for x, y in zip(a, b)])),
           "pos_frac_rms_diff": float(np.sqrt(np.mean([(x["pos_frac"] - y["pos_frac"]) ** 2 for x, y in zip(a, b)]))),
           "score_max_rel_rms": float(np.sqrt(np.mean([((x["score_max"] - y["score_max"]) / max(abs(y["score_max"]), 1e-9)) ** 2 for x, y in zip(a, b)])))}
    for t in score_thresholds:
        rep[f"top_score_gt_{t:g}_same"] = float(np.mean([(x["top_score"] > t) == (y["top_score"] > t) for x, y in zip(a, b)]))
    for t in cue_thresholds:
        rep[f"score_max_gt_{t:.3g}_same"] = float(np.mean([(x["score_max"] > t) == (y["score_max"] > t) for x, y in zip(a, b)]))
    return rep
