#!/usr/bin/env python
"""Micro-benchmark of the fused attention kernels through the C-ABI (shapes of the three towers, B=32)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vstar_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
P = lambda t: ctypes.c_void_p(t.data_ptr())
B = int(os.environ.get("B", "32"))
print(f"{'tower':<14s} {'S':>5s} {'H':>3s} {'D':>4s} {'ms':>8s} {'TFLOP/s':>8s}")
for name, S, H, D, causal, theta in [("owl-vit", 2305, 12, 64, 0, 0.0), ("clip-L@336", 577, 16, 64, 0, 0.0), ("llama S=640", 640, 32, 128, 1, 10000.0)]:
    qkv = torch.randn(B * S, 3 * H * D, device=dev).bfloat16()
    out = torch.empty(B * S, H * D, device=dev, dtype=torch.bfloat16)
    nb = lib.vstar_op_attention_workspace(B, S, H, D)
    ws = torch.zeros(nb, dtype=torch.uint8, device=dev)
    run = lambda: lib.vstar_op_attention(None, P(qkv), P(out), P(ws), nb, B, S, H, D, causal, theta)
    for _ in range(2): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 5
    e0.record()
    for _ in range(it): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / it          # includes rope + V-transpose prep and a host sync per call
    fl = 4.0 * B * H * S * S * D * (0.5 if causal else 1.0)
    print(f"{name:<14s} {S:5d} {H:3d} {D:4d} {ms:8.3f} {fl / ms / 1e9:8.1f}")
