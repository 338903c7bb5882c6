#!/usr/bin/env python
"""Kernel-level timing of the flash-attention kernel alone (no RoPE prep: theta = 0), tower shapes at B = 32.
usage: [VSTAR_LIB=...] python tools/attn_kernel_bench.py [owl|clip|llama ...]"""
import ctypes, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vstar_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
P = lambda t: ctypes.c_void_p(t.data_ptr())
B = 32
SHAPES = [("owl-vit", 2305, 12, 64, 0), ("clip-L@336", 577, 16, 64, 0), ("llama S=640 causal", 640, 32, 128, 1)]
if len(sys.argv) > 1:                                   # e.g. `attn_kernel_bench.py owl` (counter passes on one shape)
    SHAPES = [s for s in SHAPES if any(s[0].startswith(a) for a in sys.argv[1:])]
for name, S, H, D, causal in SHAPES:
    qkv = torch.randn(B * S, 3 * H * D, device=dev).bfloat16()
    out = torch.empty(B * S, H * D, device=dev, dtype=torch.bfloat16)
    nb = lib.vstar_op_attention_workspace(B, S, H, D)
    ws = torch.zeros(nb, dtype=torch.uint8, device=dev)
    run = lambda: lib.vstar_op_attention(None, P(qkv), P(out), P(ws), nb, B, S, H, D, causal, 0.0)
    for _ in range(3): run()
    torch.cuda.synchronize()
    it = 30
    t0 = time.perf_counter()
    for _ in range(it): run()          # each call synchronises: wall time / call = kernel + ~15 us of launch + sync
    ms = (time.perf_counter() - t0) / it * 1e3
    fl = 4.0 * B * H * S * S * D * (0.5 if causal else 1.0)
    print(f"{name:<20s} {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TFLOP/s")
