#!/usr/bin/env python
"""Diagnostic: GEMM time vs M around a tile boundary (K=768, N=2304)."""
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
for M in (73728, 73729, 73760, 73984, 65536, 65568, 18432, 18464):
    ms = bench(M, 2304, 768); print(f"M={M:6d} tiles_m={(M+255)//256:4d} {ms:7.3f} ms {2.0*M*2304*768/ms/1e9:7.1f} TF/s")
