#!/usr/bin/env python
"""Interleaved A/B of GEMM kernels / builds on the LLaMA shapes of the bench batch, ONE process, same operands (round 6).

usage: python tools/gemm4w_ab.py [--variants name=path/to/lib.so ...] [--iters 30] [--rounds 3] [--epilogues]
Columns: hipBLASLt (torch F.linear, measurement yardstick only), then for every library (default: the in-tree one) its 8-wave
gemm256 (VSTAR_EPI_TILE256) and its 4-wave gemm4w (VSTAR_EPI_TILE4W).  Best of `rounds` interleaved rounds, TFLOP/s on N(0,1) operands.
With --epilogues the product's fused forms are timed too (o_proj / down + residual, gate|up SiLU*up) — no library column there."""
import argparse
import ctypes
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vstar_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--variants", nargs="*", default=[])
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--epilogues", action="store_true")
ap.add_argument("--zeros", action="store_true")
ap.add_argument("--out-pad", type=int, default=0, help="output rows padded by this many elements (row pitch of C)")
ap.add_argument("--vit", action="store_true", help="the ViT shapes of the bench batch with M rounded down to whole 256-row tiles (gemm4w domain)")
ap.add_argument("--lda-pad", type=int, default=0, help="A rows padded by this many elements (row stride not a power of two)")
args = ap.parse_args()
dev = torch.device("cuda:0")
base = _lib.load()
libs = [("tree", base)]
for v in args.variants:
    name, path = v.split("=", 1)
    L = ctypes.CDLL(os.path.abspath(path))
    L.vstar_op_gemm.argtypes = base.vstar_op_gemm.argtypes
    L.vstar_op_gemm.restype = ctypes.c_int
    libs.append((name, L))
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
M = args.batch * 640
shapes = [("qkv", M, 12288, 4096, 0, 0), ("o", M, 4096, 4096, 0, 0), ("gate|up", M, 22016, 4096, 0, 0), ("down", M, 4096, 11008, 0, 0),
          ("square8k", 8192, 8192, 8192, 0, 0)]
if args.vit:
    Mc, Mo = 18464 // 256 * 256, 73760 // 256 * 256
    shapes = [("clip qkv", Mc, 3072, 1024, 0, 0), ("clip out+res", Mc, 1024, 1024, 0, 1), ("clip fc1 qgelu", Mc, 4096, 1024, 1, 0),
              ("clip fc2+res", Mc, 1024, 4096, 0, 1), ("owl qkv", Mo, 2304, 768, 0, 0), ("owl out+res", Mo, 768, 768, 0, 1),
              ("owl fc1 qgelu", Mo, 3072, 768, 1, 0), ("owl fc2+res", Mo, 768, 3072, 0, 1)]
if args.epilogues:
    shapes += [("o+res", M, 4096, 4096, 0, 1), ("gate|up silu", M, 22016, 4096, 4, 0), ("down+res", M, 4096, 11008, 0, 1)]


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


cols = ["hipblaslt"] + [f"{n}:{k}" for n, _ in libs for k in ("256", "4w")]
print(f"# {torch.cuda.get_device_name(0)}; operands {'zeros' if args.zeros else 'N(0,1) / N(0,1/K)'}; {args.iters} launches x {args.rounds} rounds, best; TFLOP/s")
print(f"{'shape':<14s} | " + " ".join(f"{c:>12s}" for c in cols))
torch.backends.cuda.preferred_blas_library("hipblaslt")
for name, Mm, N, K, epi, has_res in shapes:
    n_out = N // 2 if epi == 4 else N
    if args.zeros:
        a = torch.zeros(Mm, K + args.lda_pad, device=dev, dtype=torch.bfloat16)[:, :K]
        w = torch.zeros(N, K, device=dev, dtype=torch.bfloat16)
    else:
        a = torch.randn(Mm, K + args.lda_pad, device=dev).bfloat16()[:, :K]
        w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    lda = K + args.lda_pad
    res = torch.randn(Mm, n_out, device=dev).bfloat16() if has_res else None
    ldc = n_out + args.out_pad
    c = torch.empty(Mm, ldc, device=dev, dtype=torch.bfloat16)
    fns = {}
    if not epi and not has_res:
        fns["hipblaslt"] = lambda: F.linear(a, w)
    for ln, L in libs:
        for k, flag in (("256", _lib.EPI_TILE256), ("4w", _lib.EPI_TILE4W)):
            def f(L=L, flag=flag):
                rc = L.vstar_op_gemm(None, P(a), lda, P(w), None, P(res), n_out, P(c), ldc, 0, Mm, N, K, epi | 0x100 | flag)
                assert rc == 0, rc
            fns[f"{ln}:{k}"] = f
    best = {k: 1e9 for k in fns}
    for _ in range(args.rounds):
        for k, f in fns.items():
            best[k] = min(best[k], timed(f, args.iters))
    tf = lambda ms: 2.0 * Mm * N * K / ms / 1e9  # noqa: E731
    print(f"{name:<14s} | " + " ".join((f"{tf(best[c_]):12.0f}" if c_ in best else f"{'-':>12s}") for c_ in cols))
