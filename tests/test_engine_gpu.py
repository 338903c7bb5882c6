"""End-to-end parity of the HIP engine (through the C-ABI) on the MI355X.

(1) against the golden vectors produced by the REFERENCE's own model_forward(inference=True) (tests/golden/*.npz);
(2) against the fp32 oracle on fresh seeded inputs, batched, at both CLIP geometries.

Tolerance: north_star asks for 1e-3 relative against the reference's PyTorch path.  The reference runs bf16 end to end
and so does the engine (bf16 storage, fp32 accumulate, the reference's rounding points); two bf16 evaluations of the same
graph differ by the bf16 rounding noise accumulated over the depth of the network.  That noise is MEASURED here, per
output, as the distance between the reference algorithm evaluated in bf16 on torch-CPU (the oracle with a bf16 state
dict) and the fp32 golden vectors: 6e-3 (CLIP features) .. 1.7e-2 (logits) .. 4e-2 (masks) relative L2 on these fixtures.
Gate per output/tap:  rel_L2(engine, golden) <= max(2e-2, 3 x rel_L2(bf16 reference algorithm, golden)),
plus sigmoid box outputs <= 1e-2 absolute and an identical arg-max box.  Measured values are printed.
The 1e-3 gate is applied where it is meaningful: the fp32-output kernels in tests/test_ops_gpu.py.
"""
import glob
import os

import numpy as np
import pytest
import torch

from _parity import assert_mask_within_bf16_noise, assert_within_bf16_noise, fmt
from oracle import vsm_oracle
from oracle.gen_golden import make_inputs
from vstar_amd.config import VSMConfig
from vstar_amd.engine import VstarEngine, loc_positions
from vstar_amd.weights import random_state_dict

pytestmark = pytest.mark.gpu
GOLDEN = sorted(p for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "tiny*.npz")) if not p.endswith("_bf16.npz"))


def rel_l2(got, ref):
    got = np.asarray(got, dtype=np.float64).reshape(-1)
    ref = np.asarray(ref, dtype=np.float64).reshape(-1)
    return float(np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-30))


_ENGINES = {}


def engine_for(cfg, wseed, fold="1"):
    """fold: VSTAR_FOLD_NORMS at vstar_create — "1" (default) folds the LLaMA RMSNorms into the consuming linears, "0" keeps the
    reference's rounding points (normalised x rounded to bf16, then multiplied by w and rounded again)."""
    key = (cfg.clip_image_size, wseed, fold)
    if key not in _ENGINES:
        old = os.environ.get("VSTAR_FOLD_NORMS")
        os.environ["VSTAR_FOLD_NORMS"] = fold
        try:
            eng = VstarEngine(cfg, 0)
        finally:
            if old is None:
                del os.environ["VSTAR_FOLD_NORMS"]
            else:
                os.environ["VSTAR_FOLD_NORMS"] = old
        eng.load_state_dict(random_state_dict(cfg, seed=wseed, dtype=torch.bfloat16))
        _ENGINES[key] = eng
    return _ENGINES[key]


def margin_aware_topk_equal(got, ref, k, noise_abs):
    """Top-k ORDER of `got` equals that of `ref`, except that neighbours whose reference values are closer than `noise_abs`
    (the measured bf16 noise of this output) may swap.  Returns (ok, message)."""
    got, ref = np.asarray(got, np.float64).ravel(), np.asarray(ref, np.float64).ravel()
    order_ref = np.argsort(-ref)[:k + 1]
    order_got = np.argsort(-got)[:k]
    for r in range(k):
        if order_got[r] == order_ref[r]:
            continue
        # a swap is legitimate only among reference entries within the noise of each other
        if abs(ref[order_got[r]] - ref[order_ref[r]]) > noise_abs:
            return False, f"rank {r}: got index {order_got[r]} (ref value {ref[order_got[r]]:.5f}) vs {order_ref[r]} ({ref[order_ref[r]]:.5f}), noise {noise_abs:.2e}"
    return True, ""


@pytest.mark.parametrize("fold", ["1", "0"], ids=["fold_norms", "unfolded_norms"])
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_engine_matches_reference_golden(cuda, path, fold):
    """Gate: per tap, the engine's distance from the reference's fp32 output must not exceed 1.5 x the
    distance of the REFERENCE ITSELF evaluated in bf16 (tests/golden/*_bf16.npz, oracle/gen_golden_bf16.py) from its own fp32
    output — no fixed floor; errors pooled (RMS) over the crops of a fixture.

    The mask needs care (profiles/r02_mask_head_error_budget.txt, tools/mask_head_probe.py): mask(p) = hyper . upscaled(p) and
    upscaled(p) = mu + r(p) with |mu| ~ 5 |r| (post-GELU features share a common mode), so the mask is a large, partly
    cancelling OFFSET hyper.mu plus a PATTERN hyper.r(p).  Its plain rel-L2 is heavy-tailed per crop (0.7e-2 .. 7e-2 for the
    reference's own bf16 run across seeds) because the offset's error is one scalar draw divided by a small norm.  So the mask
    is gated in its two parts: the pattern (mean removed) at the same 1.5 x rule, and the offset against the 3-sigma band of a
    noise model fed ONLY with the reference's measured bf16 noise:  sigma = sqrt(eps_h^2 + eps_mu^2) |h| |mu| / sqrt(32)
    (eps_* = rel. error of the reference-bf16 operands; random direction in 32 dims).  The reference's own bf16 offsets are
    checked against the same band, which validates the model.  The two operands are gated as taps of their own.

    Both settings of VSTAR_FOLD_NORMS run here: the folded form is the default; the unfolded form — the reference's
    rounding points, also what the decode runner and the gathered last-block rows use — must hold the same gates."""
    z = np.load(path)
    zb = np.load(path[:-4] + "_bf16.npz")
    kw = {str(k): int(v) for k, v in zip(z["cfg_keys"], z["cfg_vals"])}
    cfg = VSMConfig.tiny(**kw)
    wseed, loc_id = int(z["weight_seed"]), int(z["loc_id"])
    assert int(zb["weight_seed"]) == wseed and np.array_equal(zb["crops"], z["crops"])
    eng = engine_for(cfg, wseed, fold)
    P = cfg.n_img_tokens
    n = len(z["crops"])
    names = ("clip_features", "llm_hidden_loc", "embed_det", "embed_seg", "pred_logits", "pred_boxes", "sam_hyper",
             "sam_upscaled_mean", "mask_pattern")
    e_eng = {k: np.zeros(n) for k in names + ("low_res_masks",)}
    e_ref = {k: np.zeros(n) for k in names + ("low_res_masks",)}
    off_eng, off_ref, off_sigma = np.zeros(n), np.zeros(n), np.zeros(n)
    for i, (seed, L, img_col, loc_col) in enumerate(z["crops"]):
        clip, owl, ids = make_inputs(cfg, int(seed), int(L), int(img_col), int(loc_col), loc_id)
        loc = loc_positions(ids.numpy(), loc_id, P)
        out = eng.score_batch(clip, owl, ids.numpy(), loc)
        H = cfg.llm_hidden
        taps = {
            "clip_features": eng.debug_read("clip_features", (P + 1) * cfg.clip_hidden).reshape(P + 1, -1)[1:],
            "llm_hidden_loc": eng.debug_read("llm_hidden_loc", H),
            "embed_det": eng.debug_read("embed_det", cfg.owl_query_dim),
            "embed_seg": eng.debug_read("embed_seg", 256),
            "pred_logits": out["pred_logits"][0, :, 0],
            "pred_boxes": out["pred_boxes"][0],
            "low_res_masks": out["low_res_masks"][0, 0],
            "sam_hyper": eng.debug_read("sam_hyper", 32),
            "sam_upscaled_mean": eng.debug_read("sam_c2", 192 * 192 * 32).reshape(-1, 32).astype(np.float64).mean(axis=0),
        }
        gold, goldb = {k: z[k][i] for k in taps}, {k: zb[k][i] for k in taps}
        for d in (taps, gold, goldb):
            m = np.asarray(d["low_res_masks"], np.float64)
            d["mask_pattern"] = m - m.mean()
        for k in names + ("low_res_masks",):
            assert np.isfinite(taps[k]).all(), k
            e_eng[k][i] = rel_l2(taps[k], gold[k])
            e_ref[k][i] = rel_l2(goldb[k], gold[k])
        h, mu = gold["sam_hyper"].astype(np.float64), gold["sam_upscaled_mean"].astype(np.float64)
        off_sigma[i] = np.hypot(e_ref["sam_hyper"][i], e_ref["sam_upscaled_mean"][i]) * np.linalg.norm(h) * np.linalg.norm(mu) / np.sqrt(32.0)
        off_eng[i] = float(np.mean(taps["low_res_masks"], dtype=np.float64) - np.mean(gold["low_res_masks"], dtype=np.float64))
        off_ref[i] = float(np.mean(goldb["low_res_masks"], dtype=np.float64) - np.mean(gold["low_res_masks"], dtype=np.float64))
        assert np.abs(taps["pred_boxes"] - z["pred_boxes"][i]).max() < 1e-2
        # identical arg-max box and top-5 order, up to swaps of entries the reference's own bf16 run cannot separate
        noise_abs = 2.0 * float(np.abs(zb["pred_logits"][i] - z["pred_logits"][i]).max())
        ok, msg = margin_aware_topk_equal(taps["pred_logits"], z["pred_logits"][i], 5, noise_abs)
        assert ok, msg
        assert int(np.argmax(taps["pred_logits"])) == int(np.argmax(z["pred_logits"][i])) or noise_abs > 0 and \
            abs(np.sort(z["pred_logits"][i])[-1] - np.sort(z["pred_logits"][i])[-2]) <= noise_abs
    print("\nrel-L2 vs the reference's fp32 output:   engine (per crop)  |  reference in bf16 (per crop)  |  pooled ratio")
    rms = lambda v: float(np.sqrt(np.mean(np.square(v))))  # noqa: E731
    for k in names + ("low_res_masks",):
        ratio = rms(e_eng[k]) / rms(e_ref[k])
        print(f"  {k:<18s} {' '.join('%.2e' % v for v in e_eng[k])}  |  {' '.join('%.2e' % v for v in e_ref[k])}  |  {ratio:.2f}"
              + ("   (informational: heavy-tailed, gated as pattern + offset)" if k == "low_res_masks" else ""))
    print("  mask offset / sigma  engine " + " ".join("%.2f" % abs(a / s) for a, s in zip(off_eng, off_sigma)) +
          "  |  reference in bf16 " + " ".join("%.2f" % abs(a / s) for a, s in zip(off_ref, off_sigma)))
    for k in names:
        assert rms(e_eng[k]) <= 1.5 * rms(e_ref[k]), (k, e_eng[k], e_ref[k])
    assert (np.abs(off_ref) <= 3.0 * off_sigma).all(), ("offset noise model does not cover the reference's own bf16 run", off_ref, off_sigma)
    assert (np.abs(off_eng) <= 3.0 * off_sigma).all(), ("mask offset outside the reference's bf16 noise band", off_eng, off_sigma)


@pytest.mark.parametrize("image_size,B,L", [(224, 3, 20), (336, 4, 27)])
def test_engine_batched_vs_oracle(cuda, image_size, B, L):
    cfg = VSMConfig.tiny(clip_image_size=image_size)
    loc_id = cfg.llm_vocab - 1
    wseed = 0 if image_size == 224 else 3
    eng = engine_for(cfg, wseed)
    sd32 = {k: v.float() for k, v in random_state_dict(cfg, seed=wseed, dtype=torch.bfloat16).items()}
    g = torch.Generator().manual_seed(100 + image_size)
    clip = torch.randn(B, 3, image_size, image_size, generator=g).bfloat16()
    owl = torch.randn(B, 3, 768, 768, generator=g).bfloat16()
    ids = torch.randint(3, loc_id - 3, (B, L), generator=g)
    ids[:, 0] = 1
    ids[:, 6] = -200
    loc_cols = [L - 3, L - 4, L - 3, L - 5][:B]
    for b in range(B):
        ids[b, loc_cols[b]] = loc_id
    P = cfg.n_img_tokens
    loc = loc_positions(ids.numpy(), loc_id, P)
    verify = np.stack([loc, loc - 1], axis=1)
    out = eng.score_batch(clip, owl, ids.numpy(), loc, verify_pos=verify)
    ref = vsm_oracle.vsm_forward(sd32, cfg, clip.float(), owl.float(), ids, loc_id,
                                 verify_pos=torch.from_numpy(verify).long())
    sd16 = random_state_dict(cfg, seed=wseed, dtype=torch.bfloat16)
    r16 = vsm_oracle.vsm_forward(sd16, cfg, clip, owl, ids, loc_id)          # the same algorithm in bf16: the noise yardstick
    rep = {}
    assert_within_bf16_noise("pred_logits", out["pred_logits"], ref["pred_logits"].numpy(), r16["pred_logits"].float().numpy(), report=rep)
    assert_within_bf16_noise("pred_boxes", out["pred_boxes"], ref["pred_boxes"].numpy(), r16["pred_boxes"].float().numpy(), report=rep)
    assert np.abs(out["pred_boxes"] - ref["pred_boxes"].numpy()).max() < 1e-2
    assert_mask_within_bf16_noise(out["low_res_masks"], ref["low_res_masks"].numpy(), r16["low_res_masks"].float().numpy(),
                                  ref["sam_taps"]["sam_hyper"].numpy(), r16["sam_taps"]["sam_hyper"].float().numpy(),
                                  ref["sam_taps"]["sam_c2"].mean(dim=1).numpy(), r16["sam_taps"]["sam_c2"].float().mean(dim=1).numpy(),
                                  report=rep)
    print("\nengine / bf16-oracle noise:", fmt(rep))
    # teacher-forcing check rows: identical arg-max, except where the oracle's own decision margin (its logit at the engine's
    # choice vs its maximum) is inside the bf16 noise of a logit row.  Noise scale: lm_head is a K=hidden dot product of a
    # hidden state carrying ~1e-2 relative bf16 noise (measured above) -> ~1e-2 x the row's logit spread.
    tl = ref["tf_logits"].numpy()
    for b in range(B):
        for v in range(verify.shape[1]):
            row, got, want = tl[b, v], int(out["tf_argmax"][b, v]), int(ref["tf_argmax"][b, v])
            if got != want:
                spread = float(row.max() - row.min())
                assert row[want] - row[got] <= 2e-2 * spread, (b, v, got, want, float(row[want] - row[got]), spread)
    # batch invariance: crop 0 alone gives bit-identical records (fixed reduction order, no split-K)
    solo = eng.score_batch(clip[:1], owl[:1], ids[:1].numpy(), loc[:1])
    assert np.array_equal(solo["pred_logits"][0], out["pred_logits"][0])
    assert np.array_equal(solo["low_res_masks"][0], out["low_res_masks"][0])


def test_upsample_mask_matches_oracle(cuda):
    cfg = VSMConfig.tiny()
    eng = engine_for(cfg, 0)
    g = torch.Generator().manual_seed(9)
    low = torch.randn(1, 1, 192, 192, generator=g)
    for (h, w) in [(192, 192), (540, 960), (37, 411), (1080, 1920)]:
        got = eng.upsample_mask(low.numpy(), h, w)
        ref = vsm_oracle.upsample_mask(low, (h, w))[0, 0].numpy()
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() < 1e-4, (h, w)


def test_errors_are_reported(cuda):
    cfg = VSMConfig.tiny()
    eng = engine_for(cfg, 0)
    from vstar_amd._lib import VstarError
    clip = torch.zeros(1, 3, 224, 224).bfloat16()
    owl = torch.zeros(1, 3, 768, 768).bfloat16()
    ids = np.ones((1, 10), dtype=np.int32)      # no image token
    with pytest.raises(VstarError, match="-200"):
        eng.score_batch(clip, owl, ids, np.array([3], dtype=np.int32))
    with pytest.raises(IndexError):
        loc_positions(ids, 999, 256)


def test_engine_real_widths_vs_oracle(cuda):
    """Real 7B / CLIP-L / OWL-ViT-B WIDTHS and token counts (hidden 4096/1024/768, 32/16/12 heads, mlp 11008, S=640,
    N=577/2305) with few layers, so that every kernel runs at the exact shapes of the benchmark and is checked against the
    fp32 oracle; depth (32/23/12 layers) only repeats these shapes."""
    cfg = VSMConfig.seal_7b(336, clip_layers=3, llm_layers=2, owl_layers=2, llm_vocab=4096, max_batch=2, max_text_len=65)
    loc_id = cfg.llm_vocab - 1
    sd = random_state_dict(cfg, seed=11, dtype=torch.bfloat16)
    eng = VstarEngine(cfg, 0)
    eng.load_state_dict(sd)
    B, L = 2, 65
    g = torch.Generator().manual_seed(4)
    clip = torch.randn(B, 3, 336, 336, generator=g).bfloat16()
    owl = torch.randn(B, 3, 768, 768, generator=g).bfloat16()
    ids = torch.randint(3, loc_id - 3, (B, L), generator=g)
    ids[:, 0] = 1
    ids[:, 35] = -200
    ids[:, L - 3] = loc_id
    loc = loc_positions(ids.numpy(), loc_id, cfg.n_img_tokens)
    verify = np.stack([loc, loc + 1], axis=1)
    out = eng.score_batch(clip, owl, ids.numpy(), loc, verify_pos=verify)
    sd32 = {k: v.float() for k, v in sd.items()}
    ref = vsm_oracle.vsm_forward(sd32, cfg, clip.float(), owl.float(), ids, loc_id, verify_pos=torch.from_numpy(verify).long())
    P = cfg.n_img_tokens
    taps = {
        "clip_features": eng.debug_read("clip_features", B * (P + 1) * cfg.clip_hidden).reshape(B, P + 1, -1)[:, 1:],
        "llm_hidden_loc": eng.debug_read("llm_hidden_loc", B * cfg.llm_hidden).reshape(B, -1),
        "embed_det": eng.debug_read("embed_det", B * 512).reshape(B, -1),
        "pred_logits": out["pred_logits"], "pred_boxes": out["pred_boxes"], "low_res_masks": out["low_res_masks"],
    }
    r16 = vsm_oracle.vsm_forward(sd, cfg, clip, owl, ids, loc_id)           # the same algorithm in bf16 on torch-CPU
    rep = {}
    for k, v in taps.items():
        if k != "low_res_masks":
            assert_within_bf16_noise(k, v, ref[k].numpy(), r16[k].float().numpy(), report=rep)
    assert_mask_within_bf16_noise(out["low_res_masks"], ref["low_res_masks"].numpy(), r16["low_res_masks"].float().numpy(),
                                  ref["sam_taps"]["sam_hyper"].numpy(), r16["sam_taps"]["sam_hyper"].float().numpy(),
                                  ref["sam_taps"]["sam_c2"].mean(dim=1).numpy(), r16["sam_taps"]["sam_c2"].float().mean(dim=1).numpy(),
                                  report=rep)
    print("\nreal-width engine / bf16-oracle noise (rel-L2 vs fp32 oracle):", fmt(rep))
    assert np.abs(out["pred_boxes"] - ref["pred_boxes"].numpy()).max() < 2e-2
    # teacher-forced arg-max at the verify rows: equal unless the oracle's own margin is inside the logit noise
    tl = ref["tf_logits"].numpy()
    for b in range(B):
        for v in range(2):
            got, want = int(out["tf_argmax"][b, v]), int(ref["tf_argmax"][b, v])
            if got != want:
                assert tl[b, v, want] - tl[b, v, got] <= 2e-2 * float(tl[b, v].max() - tl[b, v].min()), (b, v, got, want)
    eng.close()


def test_prefill_argmax_equals_first_decoded_token_at_real_widths(cuda):
    """the scoring prefill (RMSNorms folded into the linears, rstd applied to the fp32 accumulators) and the KV-cached
    decode runner (normalise, round, then the linears with the same folded weights) no longer share rounding points, yet
    `VSM._decode_fallback` relies on them agreeing on arg-max tokens.  At the real LLaMA widths: the arg-max the teacher-forced
    prefill reports at the LAST prompt position must be the first token `vstar_vsm_generate` emits, for several prompts — unless
    the top-2 logits are nearly tied (the fp32 oracle's own margin decides)."""
    cfg = VSMConfig.seal_7b(224, clip_layers=2, llm_layers=3, owl_layers=1, llm_vocab=4096, max_batch=2, max_text_len=80)
    sd = random_state_dict(cfg, seed=13, dtype=torch.bfloat16)
    eng = VstarEngine(cfg, 0)
    eng.load_state_dict(sd)
    P = cfg.n_img_tokens
    g = torch.Generator().manual_seed(8)
    sd32 = {k: v.float() for k, v in sd.items()}
    agree = 0
    for trial in range(4):
        L = 30 + 7 * trial
        clip = torch.randn(1, 3, 224, 224, generator=g).bfloat16()
        ids = torch.randint(3, cfg.llm_vocab - 3, (1, L), generator=g)
        ids[0, 0] = 1
        ids[0, 12] = -200
        last = L - 1 + (P - 1)
        out = eng.score_batch(clip, None, ids.numpy(), np.asarray([last], np.int32), verify_pos=np.asarray([[last]], np.int32), skip_owl=True)
        tf = int(out["tf_argmax"][0, 0])
        gen = eng.generate(clip, [int(t) for t in ids[0]], 1, -1)
        assert len(gen) == 1
        if gen[0] == tf:
            agree += 1
            continue
        logits = vsm_oracle.greedy_next_logits(sd32, cfg, clip.float(), ids)[0]
        span = float(logits.max() - logits.min())
        assert abs(float(logits[gen[0]] - logits[tf])) <= 0.02 * span, (trial, gen[0], tf)      # a near-tie may flip; nothing else may
    print(f"\nprefill arg-max == first decoded token on {agree}/4 prompts")
    assert agree >= 3
    eng.close()


def test_decode_tile_major_weights_are_bit_identical(cuda, monkeypatch):
    """Round 5: the decode GEMV streams q|k|v, gate|up and down from a TILE-MAJOR copy of the weights (GemmParams::W_tiled: one
    sequential HBM region per workgroup instead of 16 - 32 row streams).  Same requests, same LDS image, same arithmetic: the greedy
    continuation at the real LLaMA widths (where the LDS-ring GEMV runs) equals the row-major one (VSTAR_DECODE_TILED=0) token for
    token."""
    cfg = VSMConfig.seal_7b(224, clip_layers=2, llm_layers=3, owl_layers=1, llm_vocab=4096, max_batch=2, max_text_len=80)
    sd = random_state_dict(cfg, seed=13, dtype=torch.bfloat16)
    g = torch.Generator().manual_seed(21)
    clip = torch.randn(1, 3, 224, 224, generator=g).bfloat16()
    ids = torch.randint(3, cfg.llm_vocab - 3, (1, 40), generator=g)
    ids[0, 0] = 1
    ids[0, 12] = -200
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("VSTAR_DECODE_TILED", flag)          # read when the decode runner is built (first generate call)
        eng = VstarEngine(cfg, 0)
        eng.load_state_dict(sd)
        outs.append(eng.generate(clip, [int(t) for t in ids[0]], 16, -1))
        eng.close()
    assert len(outs[0]) == 16 and outs[0] == outs[1], outs


def test_fused_rope_epilogue_is_bit_identical_to_separate_pass(cuda, monkeypatch):
    """The q|k RoPE fused into the 256^2 GEMM epilogue (M >= 1024 rows) == GEMM followed by rope_kernel, bit for bit."""
    cfg = VSMConfig.tiny()
    loc_id = cfg.llm_vocab - 1
    sd = random_state_dict(cfg, seed=0, dtype=torch.bfloat16)
    B, L = 4, 24                       # 4 x (256 + 23) = 1116 rows >= 1024: the 256^2 kernel takes the qkv projection
    g = torch.Generator().manual_seed(77)
    clip = torch.randn(B, 3, 224, 224, generator=g).bfloat16()
    owl = torch.randn(B, 3, 768, 768, generator=g).bfloat16()
    ids = torch.randint(3, loc_id - 3, (B, L), generator=g)
    ids[:, 0] = 1
    ids[:, 5] = -200
    ids[:, L - 3] = loc_id
    loc = loc_positions(ids.numpy(), loc_id, cfg.n_img_tokens)
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("VSTAR_FUSED_ROPE", flag)
        eng = VstarEngine(cfg, 0)
        eng.load_state_dict(sd)
        outs.append(eng.score_batch(clip, owl, ids.numpy(), loc))
        del eng
    for k in ("pred_logits", "pred_boxes", "low_res_masks"):
        assert np.array_equal(outs[0][k], outs[1][k]), k


def test_w8a8_mode_matches_fake_quant_oracle(cuda):
    """BASELINE config 5 (fp8 weights / fp8 MFMA): with llm_w8a8=1 and >= 1024 rows the LLaMA linears run W8A8.  Parity target
    = the oracle with the same fake quantisation (per-token / per-output-channel absmax/448 e4m3) evaluated in bf16; the
    bf16 engine is compared too, to show what the quantisation itself costs."""
    base = VSMConfig.tiny()
    cfg8 = VSMConfig.tiny(llm_w8a8=1)
    loc_id = base.llm_vocab - 1
    sd = random_state_dict(base, seed=0, dtype=torch.bfloat16)
    B, L = 4, 24
    g = torch.Generator().manual_seed(78)
    clip = torch.randn(B, 3, 224, 224, generator=g).bfloat16()
    owl = torch.randn(B, 3, 768, 768, generator=g).bfloat16()
    ids = torch.randint(3, loc_id - 3, (B, L), generator=g)
    ids[:, 0] = 1
    ids[:, 5] = -200
    ids[:, L - 3] = loc_id
    loc = loc_positions(ids.numpy(), loc_id, base.n_img_tokens)
    e8 = VstarEngine(cfg8, 0)
    e8.load_state_dict(sd)
    out8 = e8.score_batch(clip, owl, ids.numpy(), loc)
    h8 = e8.debug_read("llm_hidden_loc", B * base.llm_hidden).reshape(B, -1)
    assert not e8.w8a8_mx_active()                                     # 4 x 279 rows: not a multiple of 256 -> the per-token scheme
    e16 = engine_for(base, 0)
    out16 = e16.score_batch(clip, owl, ids.numpy(), loc)
    h16 = e16.debug_read("llm_hidden_loc", B * base.llm_hidden).reshape(B, -1)
    ref8 = vsm_oracle.vsm_forward(sd, cfg8, clip, owl, ids, loc_id)              # bf16 arithmetic + fake quant
    ref16 = vsm_oracle.vsm_forward(sd, base, clip, owl, ids, loc_id)
    rep = {
        "hidden: engine8 vs oracle8": rel_l2(h8, ref8["llm_hidden_loc"].float().numpy()),
        "hidden: engine16 vs oracle16": rel_l2(h16, ref16["llm_hidden_loc"].float().numpy()),
        "hidden: engine8 vs engine16": rel_l2(h8, h16),
        "logits: engine8 vs oracle8": rel_l2(out8["pred_logits"], ref8["pred_logits"].float().numpy()),
        "logits: engine8 vs engine16": rel_l2(out8["pred_logits"], out16["pred_logits"]),
        "masks: engine8 vs oracle8": rel_l2(out8["low_res_masks"], ref8["low_res_masks"].float().numpy()),
    }
    print({k: "%.2e" % v for k, v in rep.items()})
    assert rep["hidden: engine8 vs engine16"] > 1e-4                   # the fp8 path really ran
    noise = max(rep["hidden: engine16 vs oracle16"], 5e-3)
    assert rep["hidden: engine8 vs oracle8"] <= 6 * noise               # same order as the bf16 path's own noise
    assert rep["hidden: engine8 vs engine16"] <= 0.15                  # and the quantisation costs a few percent
    assert rep["logits: engine8 vs oracle8"] <= 6e-2 and rep["masks: engine8 vs oracle8"] <= 8e-2


def test_w8a8_whole_chain_block_scaled_in_a_process_of_its_own(cuda):
    """VSTAR_W8A8_MX=2 (opt-in: the residual stream block-scaled too, RMSNorms folded) is read once per process: the real-width test
    below in a subprocess with the level set — engine vs the oracle's restatement of THAT scheme (linear_w8a8_mx_folded)."""
    import subprocess
    import sys
    env = dict(os.environ, VSTAR_W8A8_MX="2")
    r = subprocess.run([sys.executable, "-m", "pytest", __file__, "-q", "-m", "gpu", "-k", "test_w8a8_real_widths", "-p", "no:cacheprovider"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "1 passed" in r.stdout, r.stdout[-2000:] + r.stderr[-1000:]


def test_w8a8_real_widths(cuda):
    """W8A8 at the real 7B widths (K = 4096 / 11008, S = 640; 3 LLaMA layers): engine vs the fake-quant oracle, and what
    the quantisation costs against the bf16 engine on the same inputs."""
    kw = dict(clip_layers=3, llm_layers=3, owl_layers=2, llm_vocab=4096, max_batch=2, max_text_len=65)
    cfg16, cfg8 = VSMConfig.seal_7b(336, **kw), VSMConfig.seal_7b(336, llm_w8a8=1, **kw)
    loc_id = cfg16.llm_vocab - 1
    sd = random_state_dict(cfg16, seed=11, dtype=torch.bfloat16)
    B, L = 2, 65
    g = torch.Generator().manual_seed(4)
    clip = torch.randn(B, 3, 336, 336, generator=g).bfloat16()
    owl = torch.randn(B, 3, 768, 768, generator=g).bfloat16()
    ids = torch.randint(3, loc_id - 3, (B, L), generator=g)
    ids[:, 0] = 1
    ids[:, 35] = -200
    ids[:, L - 3] = loc_id
    loc = loc_positions(ids.numpy(), loc_id, cfg16.n_img_tokens)
    res = {}
    for name, cfg in (("bf16", cfg16), ("w8a8", cfg8)):
        eng = VstarEngine(cfg, 0)
        eng.load_state_dict(sd)
        out = eng.score_batch(clip, owl, ids.numpy(), loc)
        res[name] = (eng.debug_read("llm_hidden_loc", B * cfg.llm_hidden).reshape(B, -1), out)
        mx = eng.w8a8_mx_active()
        eng.close()
    assert mx == int(os.environ.get("VSTAR_W8A8_MX", "1"))              # 2 x 640 rows: block-scaled o_proj / down_proj inputs (round 6; 2 = the whole chain, opt-in)
    sd32 = {k: v.float() for k, v in sd.items()}
    ref8 = vsm_oracle.vsm_forward(sd32, cfg8, clip.float(), owl.float(), ids, loc_id, w8a8_mx=mx)
    rep = {"hidden w8a8 vs fake-quant oracle": rel_l2(res["w8a8"][0], ref8["llm_hidden_loc"].numpy()),
           "hidden w8a8 vs bf16 engine": rel_l2(res["w8a8"][0], res["bf16"][0]),
           "logits w8a8 vs fake-quant oracle": rel_l2(res["w8a8"][1]["pred_logits"], ref8["pred_logits"].numpy()),
           "logits w8a8 vs bf16 engine": rel_l2(res["w8a8"][1]["pred_logits"], res["bf16"][1]["pred_logits"]),
           "masks w8a8 vs bf16 engine": rel_l2(res["w8a8"][1]["low_res_masks"], res["bf16"][1]["low_res_masks"])}
    print("\nreal-width W8A8:", {k: f"{v:.2e}" for k, v in rep.items()})
    assert rep["hidden w8a8 vs bf16 engine"] > 1e-4
    # e4m3 keeps 3 mantissa bits: every W8A8 linear carries ~3-4 % of output-relative noise on random data (it does not
    # average out with K: signal and noise both grow like sqrt(K)), and which code an activation rounds to depends on
    # bf16-level upstream differences, so engine and oracle agree to the same order as the quantisation noise itself
    assert rep["hidden w8a8 vs fake-quant oracle"] < 9e-2 and rep["logits w8a8 vs fake-quant oracle"] < 3e-2
    assert rep["hidden w8a8 vs bf16 engine"] < 0.12
