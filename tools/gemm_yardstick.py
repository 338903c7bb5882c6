#!/usr/bin/env python
"""Outside yardstick for the bf16 GEMM family: the vendor libraries behind torch.nn.functional.linear
(hipBLASLt, then rocBLAS) against this repository's gemm256 through the C-ABI, on the SAME box, SAME shapes, SAME random operands,
interleaved in one process (library, engine, library, engine ...) so that clock / thermal drift hits both alike.

MEASUREMENT TOOL ONLY — nothing under vstar_amd/ links or calls a vendor BLAS (grep csrc/: clean); this script is the one place
where torch.matmul appears next to the engine.

usage: python tools/gemm_yardstick.py [--batch 32] [--iters 30] [--rounds 3] > profiles/r05_gemm_yardstick.txt
Plain GEMM only (no bias / residual / activation on either side): C[M,N] = A[M,K] . W[N,K]^T, bf16 in, fp32 accumulate, bf16 out.
"""
import argparse
import ctypes
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vstar_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--zeros", action="store_true", help="zero-filled operands (the power-unconstrained figure; never the headline)")
args = ap.parse_args()
lib = _lib.load()
dev = torch.device("cuda:0")
B = args.batch
S, Nc, No = 640, 577, 2305
shapes = [("llama qkv", B * S, 12288, 4096), ("llama o", B * S, 4096, 4096), ("llama gate|up", B * S, 22016, 4096),
          ("llama down", B * S, 4096, 11008), ("clip qkv", B * Nc, 3072, 1024), ("clip out", B * Nc, 1024, 1024),
          ("clip fc1", B * Nc, 4096, 1024), ("clip fc2", B * Nc, 1024, 4096), ("owl qkv", B * No, 2304, 768),
          ("owl out", B * No, 768, 768), ("owl fc1", B * No, 3072, 768), ("owl fc2", B * No, 768, 3072),
          ("square 8192", 8192, 8192, 8192)]
P = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731


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


backends = []
for name in ("hipblaslt", "rocblas"):
    try:
        torch.backends.cuda.preferred_blas_library(name)
        backends.append(name)
    except Exception as exc:  # noqa: BLE001
        print(f"# backend {name} not selectable: {exc}")
print(f"# torch {torch.__version__}, device {torch.cuda.get_device_name(0)}, operands {'zeros' if args.zeros else 'N(0,1) / N(0,1/K)'}, "
      f"{args.iters} launches x {args.rounds} interleaved rounds, best round per column; TFLOP/s = 2MNK / time")
print(f"{'shape':<14s} {'M':>7s} {'N':>6s} {'K':>6s} | " + " ".join(f"{b + ' ms':>13s} {'TF/s':>7s}" for b in backends) +
      f" | {'gemm256 ms':>11s} {'TF/s':>7s} | {'gemm4w ms':>10s} {'TF/s':>7s} | gemm256 / lib   gemm4w / lib")
for name, M, N, K in shapes:
    if args.zeros:
        a = torch.zeros(M, K, device=dev, dtype=torch.bfloat16)
        w = torch.zeros((N + 255) // 256 * 256, K, device=dev, dtype=torch.bfloat16)
    else:
        a = torch.randn(M, K, device=dev).bfloat16()
        w = torch.zeros((N + 255) // 256 * 256, K, device=dev, dtype=torch.bfloat16)
        w[:N] = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    wl = w[:N]
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)

    def eng():
        rc = lib.vstar_op_gemm(None, P(a), K, P(w), None, None, N, P(c), N, 0, M, N, K, 0 | 0x100)
        assert rc == 0

    def eng4():
        rc = lib.vstar_op_gemm(None, P(a), K, P(w), None, None, N, P(c), N, 0, M, N, K, 0 | 0x100 | _lib.EPI_TILE4W)
        assert rc == 0

    has4 = M % 256 == 0 and N % 256 == 0 and K % 128 == 0        # gemm4w's domain (round 6)
    best = {b: 1e9 for b in backends}
    best_e = best_4 = 1e9
    for _ in range(args.rounds):
        for b in backends:
            torch.backends.cuda.preferred_blas_library(b)
            best[b] = min(best[b], timed(lambda: F.linear(a, wl), args.iters))
        best_e = min(best_e, timed(lambda: (lib.vstar_op_gemm(None, P(a), K, P(w), None, None, N, P(c), N, 0, M, N, K, 0 | 0x100 | _lib.EPI_TILE256)
                                            if M >= 1024 and K % 128 == 0 else eng()), args.iters))
        if has4:
            best_4 = min(best_4, timed(eng4, args.iters))
    # same numbers: the library's output is the check of the engine's (bf16 rounding of an fp32 accumulation either way)
    torch.backends.cuda.preferred_blas_library(backends[0])
    ref = F.linear(a, wl).float()
    eng()
    err = float((c.float() - ref).norm() / ref.norm().clamp_min(1e-30))
    tf = lambda ms: 2.0 * M * N * K / ms / 1e9  # noqa: E731
    lib_best = min(best.values())
    if has4:
        c4 = torch.empty_like(c)
        rc = lib.vstar_op_gemm(None, P(a), K, P(w), None, None, N, P(c4), N, 0, M, N, K, _lib.EPI_TILE4W)
        assert rc == 0 and torch.equal(c4, c), "gemm4w differs from gemm256"
    print(f"{name:<14s} {M:7d} {N:6d} {K:6d} | " + " ".join(f"{best[b]:13.3f} {tf(best[b]):7.0f}" for b in backends) +
          f" | {best_e:11.3f} {tf(best_e):7.0f} | " + (f"{best_4:10.3f} {tf(best_4):7.0f}" if has4 else f"{'-':>10s} {'-':>7s}") +
          f" | x{lib_best / best_e:.3f}   " + (f"x{lib_best / best_4:.3f}" if has4 else "-") + f"   (rel diff of outputs {err:.1e})")
