#!/usr/bin/env python
"""Does a power-of-two row pitch of A / C cost the 256^2 GEMM anything (HBM / L2 channel aliasing)?  Same shape with dense and
padded leading dimensions.  usage: python tools/gemm_ld_probe.py"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vstar_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
print(f"{'shape':<26s} {'lda':>6s} {'ldc':>6s} {'ms':>8s} {'TFLOP/s':>8s}")
for name, M, N, K, epi, has_res in [("llama o (no res)", 20480, 4096, 4096, 0, 0), ("llama o +res", 20480, 4096, 4096, 0, 1),
                                    ("llama qkv", 20480, 12288, 4096, 0, 0), ("llama down +res", 20480, 4096, 11008, 0, 1),
                                    ("llama gate_up", 20480, 22016, 4096, 4, 0)]:
    for pa, pc in [(0, 0), (64, 0), (0, 64), (64, 64), (0, 0)]:
        lda, n_out = K + pa, (N // 2 if epi == 4 else N)
        ldc = n_out + pc
        a = torch.randn(M, lda, device=dev).bfloat16()
        w = (torch.randn((N + 255) // 256 * 256, K, device=dev) / K ** 0.5).bfloat16()
        c = torch.empty(M, ldc, device=dev, dtype=torch.bfloat16)
        res = torch.randn(M, ldc, device=dev).bfloat16() if has_res else None
        run = lambda: lib.vstar_op_gemm(None, P(a), lda, P(w), None, P(res), ldc, P(c), ldc, 0, M, N, K, epi | 0x100)
        for _ in range(3): assert run() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 30
        print(f"{name:<26s} {lda:6d} {ldc:6d} {ms:8.3f} {2.0 * M * N * K / ms / 1e9:8.1f}")
