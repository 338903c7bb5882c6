#!/usr/bin/env python
"""Ablation of the 256x256 GEMM main loop on the LLaMA gate_up shape: full / no-MFMA / no-DMA / neither (diagnostic)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vstar_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
P = lambda t: ctypes.c_void_p(t.data_ptr())
M, N, K = 20480, 4096, 8192
a = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16(); c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
run = lambda: lib.vstar_op_gemm(None, P(a), K, P(w), None, None, 0, P(c), N, 0, M, N, K, 0x100)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
tiles_per_cu = (M // 256) * (N // 256) / 256
print(f"debug={os.environ.get('VSTAR_GEMM_DEBUG','0'):>3s}  {ms:7.3f} ms  -> {ms*1000/tiles_per_cu/(K/64):6.3f} us per K-tile per CU")
