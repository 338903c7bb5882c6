"""numpy restatement of Pillow's 8-bit antialiased bicubic resize — TEST INFRASTRUCTURE ONLY.

The reference preprocesses every crop with HF `CLIPImageProcessor` / `OwlViTProcessor` (visual_search.py:186-194), i.e.
`PIL.Image.resize(size, resample=BICUBIC)` on uint8 RGB.  Pillow (third-party, not in the reference tree; algorithm:
src/libImaging/Resample.c — `precompute_coeffs`, `normalize_coeffs_8bpc`, `ImagingResampleHorizontal_8bpc`,
`ImagingResampleVertical_8bpc`) resamples in two passes, horizontal then vertical, with per-output-pixel windows whose
support grows with the down-scale factor, coefficients quantised to 22 fractional bits, and a uint8 intermediate image.
`resize_u8` reproduces that arithmetic; tests/test_host.py pins it bit-exactly against the installed Pillow, and the HIP
kernels in vstar_amd/csrc/preprocess.hip are then checked against it / against Pillow on the GPU.
"""
from __future__ import annotations

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _bicubic(x: float) -> float:
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size: int, out_size: int):
    """Returns (bounds [out,2] = (xmin, count), int32 coefficients [out, ksize])."""
    scale = float(in_size) / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = 2.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = np.zeros(ksize, np.float64)
        ww = 0.0
        for x in range(xmax):
            w = _bicubic((x + xmin - center + 0.5) * ss)
            k[x] = w
            ww += w
        for x in range(xmax):
            if ww != 0.0:
                k[x] /= ww
        bounds[xx] = (xmin, xmax)
        for x in range(ksize):
            v = k[x] * (1 << PRECISION_BITS)
            kk[xx, x] = int(-0.5 + v) if k[x] < 0 else int(0.5 + v)
    return bounds, kk


def _pass(img: np.ndarray, out_size: int, axis: int) -> np.ndarray:
    """One resampling pass of a uint8 [H, W, C] image along `axis` (1 = horizontal, 0 = vertical)."""
    in_size = img.shape[axis]
    bounds, kk = precompute_coeffs(in_size, out_size)
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((out_size,) + src.shape[1:], np.uint8)
    for xx in range(out_size):
        xmin, cnt = bounds[xx]
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        acc += np.tensordot(kk[xx, :cnt].astype(np.int64), src[xmin:xmin + cnt], axes=(0, 0))
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_u8(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """PIL.Image.fromarray(img).resize((out_w, out_h), BICUBIC) for uint8 [H, W, 3]."""
    h, w = img.shape[:2]
    if w != out_w:
        img = _pass(img, out_w, 1)
    if h != out_h:
        img = _pass(img, out_h, 0)
    return img
