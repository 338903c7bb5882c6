#!/usr/bin/env python
"""Per-shape micro-benchmark of the bf16 MFMA GEMM through the C-ABI (random N(0,1) operands, HIP-event timing).
usage: python tools/gemm_bench.py [--batch 32] ; set VSTAR_GEMM_TILE=128 to force the 128x128 kernel."""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vstar_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--rotate", type=int, default=1, help="cycle through this many copies of W (small batches: one copy would sit in the "
                "256 MB Infinity Cache, while the engine streams every layer's own weights from HBM)")
ap.add_argument("--only", type=str, default="", help="substring filter on the shape name")
args = ap.parse_args()
lib = _lib.load()
dev = torch.device("cuda:0")
B = args.batch
S, Nc, No = 640, 577, 2305
# (name, M, N, K, epilogue, residual?) exactly as the engine launches them
shapes = [("llama qkv", B * S, 12288, 4096, 0, 0), ("llama o +res", B * S, 4096, 4096, 0, 1), ("llama o (no res)", B * S, 4096, 4096, 0, 0),
          ("llama down (no res)", B * S, 4096, 11008, 0, 0), ("llama gate_up silu", B * S, 22016, 4096, 4, 0),
          ("llama down +res", B * S, 4096, 11008, 0, 1), ("clip qkv", B * Nc, 3072, 1024, 0, 0), ("clip out +res", B * Nc, 1024, 1024, 0, 1),
          ("clip fc1 qgelu", B * Nc, 4096, 1024, 1, 0), ("clip fc2 +res", B * Nc, 1024, 4096, 0, 1), ("owl qkv", B * No, 2304, 768, 0, 0),
          ("owl out +res", B * No, 768, 768, 0, 1), ("owl fc1 qgelu", B * No, 3072, 768, 1, 0), ("owl fc2 +res", B * No, 768, 3072, 0, 1),
          ("sam conv1", B * 9216, 64, 2304, 0, 0), ("sam conv2 gelu", B * 36864, 32, 576, 2, 0), ("square 8192", 8192, 8192, 8192, 0, 0)]
P = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
print(f"{'shape':<22s} {'M':>7s} {'N':>6s} {'K':>6s} {'ms':>8s} {'TFLOP/s':>8s}")
for name, M, N, K, epi, has_res in shapes:
    if args.only and args.only not in name:
        continue
    a = torch.randn(M, K, device=dev).bfloat16()
    npad = (N + 255) // 256 * 256
    w = torch.zeros(npad, K, device=dev, dtype=torch.bfloat16)
    w[:N] = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    ws = [w] + [w.clone() for _ in range(args.rotate - 1)]
    it = [0]
    n_out = N // 2 if epi == 4 else N
    c = torch.empty(M, n_out, device=dev, dtype=torch.bfloat16)
    res = torch.randn(M, n_out, device=dev).bfloat16() if has_res else None
    bias = torch.randn(npad, device=dev).bfloat16() if name.split()[0] in ("clip", "owl", "sam") else None

    def run():
        w = ws[it[0] % len(ws)]
        it[0] += 1
        rc = lib.vstar_op_gemm(None, P(a), K, P(w), P(bias) if bias is not None else None, P(res) if res is not None else None, n_out, P(c), n_out, 0, M, N, K,
                               epi | 0x100)
        assert rc == 0

    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.iters
    print(f"{name:<22s} {M:7d} {N:6d} {K:6d} {ms:8.3f} {2.0 * M * N * K / ms / 1e9:8.1f}")
