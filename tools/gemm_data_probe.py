#!/usr/bin/env python
"""Does the DATA change the speed of the same GEMM launch?  LLaMA gate|up shape, A ~ N(0, s^2) for several s, and an A with a few
large-magnitude columns (what a raw residual stream looks like next to its RMS-normalised copy).  usage: python tools/gemm_data_probe.py"""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vstar_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
M, N, K = 20480, 22016, 4096
g = torch.Generator(device=dev).manual_seed(0)
w = (torch.randn(N, K, generator=g, device=dev) / 64).bfloat16()
c = torch.empty(M, N // 2, dtype=torch.bfloat16, device=dev)
base = torch.randn(M, K, generator=g, device=dev)
cases = {"N(0,1)": base, "N(0,8^2)": base * 8, "N(0,64^2)": base * 64, "zeros": base * 0}
spiky = base.clone(); spiky[:, ::97] *= 50
cases["N(0,1) + 1% columns x50"] = spiky
for name, a32 in cases.items():
    a = a32.bfloat16()
    run = lambda: lib.vstar_op_gemm(None, P(a), K, P(w), None, None, 0, P(c), N // 2, 0, M, N, K, _lib.EPI_SILU_MUL | _lib.EPI_NOSYNC)
    for _ in range(5): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(40): run()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 40 * 1e3
    print(f"{name:<28s} {ms:7.3f} ms  {2.0 * M * N * K / ms / 1e9:8.1f} TFLOP/s")
