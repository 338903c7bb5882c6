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
    inside the 3-sigma band of the noise model sigma = sqrt(eps_h^2 + eps_mu^2) |h| |mu| / sqrt(32) per mask (see
    tests/test_engine_gpu.py::test_engine_matches_reference_golden).  hyper*/upmean*: [B,32] operands of the final product."""
    f = lambda a, nd: np.asarray(a, np.float64).reshape((-1,) + tuple(np.asarray(a).shape[-nd:]))  # noqa: E731
    got, ref32, ref16 = f(got, 2), f(ref32, 2), f(ref16, 2)
    h32, h16, m32, m16 = f(hyper32, 1), f(hyper16, 1), f(upmean32, 1), f(upmean16, 1)
    pat = lambda m: m - m.mean(axis=(1, 2), keepdims=True)  # noqa: E731
    e, n = rel_l2(pat(got), pat(ref32)), rel_l2(pat(ref16), pat(ref32))
    if report is not None:
        report["mask_pattern"] = (e, n)
    assert e <= factor * n, f"mask pattern: engine {e:.2e} vs bf16 noise {n:.2e}"
    for b in range(got.shape[0]):
        eps = np.hypot(rel_l2(h16[b], h32[b]), rel_l2(m16[b], m32[b]))
        sigma = eps * np.linalg.norm(h32[b]) * np.linalg.norm(m32[b]) / np.sqrt(h32.shape[-1])
        off = abs(got[b].mean() - ref32[b].mean())
        off16 = abs(ref16[b].mean() - ref32[b].mean())
        if report is not None:
            report[f"mask_offset_sigma[{b}]"] = (off / sigma, off16 / sigma)
        assert off <= 3.0 * sigma, f"mask {b}: offset {off:.4f} outside 3 sigma = {3 * sigma:.4f} of the bf16 noise model"


def fmt(report):
    return "  ".join(f"{k} {a:.2e}/{b:.2e}" for k, (a, b) in report.items())
