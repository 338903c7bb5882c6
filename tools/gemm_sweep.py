#!/usr/bin/env python
"""GEMM timing sweep over tile-waves and K (diagnostic)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vstar_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
P = lambda t: ctypes.c_void_p(t.data_ptr())
def bench(M, N, K, epi=0, iters=20):
    a = torch.randn(M, K, device=dev).bfloat16(); npad = (N + 255) // 256 * 256
    w = torch.zeros(npad, K, device=dev, dtype=torch.bfloat16); w[:N] = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    run = lambda: lib.vstar_op_gemm(None, P(a), K, P(w), None, None, 0, P(c), N, 0, M, N, K, epi | 0x100)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
print("tile-waves sweep (N=4096 -> 16 n-tiles; M = 4096*w -> 256*w tiles), K=4096")
for w in (1, 2, 3, 5, 8):
    ms = bench(4096 * w, 4096, 4096); print(f"  waves={w} ms={ms:.4f} per-wave-us={ms*1000/w:.1f} TF={2*4096*w*4096*4096/ms/1e9:.0f}")
print("K sweep at exactly 1 wave of tiles (M=N=4096)")
for K in (128, 256, 512, 1024, 2048, 4096, 8192):
    ms = bench(4096, 4096, K); print(f"  K={K} ms={ms:.4f} us-per-ktile={ms*1000/(K/64):.3f}")
print("half-occupied chip: 128 tiles (M=2048,N=4096), K=4096")
ms = bench(2048, 4096, 4096); print(f"  ms={ms:.4f}")
print("16 tiles only (M=1024,N=1024), K=4096")
ms = bench(1024, 1024, 4096); print(f"  ms={ms:.4f}")
