#!/usr/bin/env python
"""Diagnostic: is the GEMM's output-store cost a bandwidth effect or a row-scatter (TLB / DRAM-page) effect?
Same tile count, same output bytes, two output layouts: wide C (row stride 6 KiB, 12 column tiles) vs one-tile-wide C
(N=256: every tile's 128 KiB of output is contiguous)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vstar_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
P = lambda t: ctypes.c_void_p(t.data_ptr())
def bench(M, N, K, iters=10):
    a = torch.randn(M, K, device=dev).bfloat16(); npad = (N + 255) // 256 * 256
    w = torch.zeros(npad, K, device=dev, dtype=torch.bfloat16); w[:N] = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    run = lambda: lib.vstar_op_gemm(None, P(a), K, P(w), None, None, 0, P(c), N, 0, M, N, K, 0x100)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for name, M, N, K in [("wide   C: 80x12 tiles, row stride 6 KiB", 20480, 3072, 4096), ("narrow C: 960x1 tiles, contiguous tiles", 245760, 256, 4096),
                      ("wide   K=768 ", 73728, 2304, 768), ("narrow K=768 ", 73728 * 9, 256, 768)]:
    ms = bench(M, N, K); print(f"{name:<44s} M={M:7d} N={N:5d} K={K:5d} {ms:8.3f} ms {2.0*M*N*K/ms/1e9:8.1f} TF/s")
