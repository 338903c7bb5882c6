#!/usr/bin/env python
"""Where an attention block's time goes (instrumented build of attention.hip, tools/experiments/attn_timeline/attention_tl.hip.txt):
per block, wave 3 accumulates wall-clock ticks (10 ns) spent in the per-tile wait+barrier, the DMA issue and the two sub-tile computes.
usage: VSTAR_LIB=vstar_amd/csrc/build/ab/lib_attntl.so python tools/experiments/attn_timeline/attn_timeline.py"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from vstar_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
P = lambda t: ctypes.c_void_p(t.data_ptr())
B = 32
for name, S, H, D, causal, slot in [("owl-vit", 2305, 12, 64, 0, 0), ("clip-L@336", 577, 16, 64, 0, 0), ("llama S=640 causal", 640, 32, 128, 1, 3)]:
    qkv = torch.randn(B * S, 3 * H * D, device=dev).bfloat16()
    out = torch.empty(B * S, H * D, device=dev, dtype=torch.bfloat16)
    nb = lib.vstar_op_attention_workspace(B, S, H, D)
    ws = torch.zeros(max(nb, 16), dtype=torch.uint8, device=dev)
    run = lambda: lib.vstar_op_attention(None, P(qkv), P(out), P(ws), nb, B, S, H, D, causal, 0.0)
    for _ in range(3): run()
    torch.cuda.synchronize()
    buf = np.zeros((4, 8), np.uint64)
    lib.vstar_debug_attn_timeline(buf.ctypes.data_as(ctypes.c_void_p), 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    lib.vstar_debug_attn_timeline(buf.ctypes.data_as(ctypes.c_void_p), 0)
    r = buf[slot].astype(np.float64)
    n = r[0]
    print(f"{name:20s} {e0.elapsed_time(e1) / 5:7.3f} ms/launch; per block (wave 3, us): total {r[1] / n / 100:7.2f} = prologue {r[5] / n / 100:5.2f} + "
          f"[wait+barrier {r[2] / n / 100:6.2f} | DMA issue {r[3] / n / 100:5.2f} | compute {r[4] / n / 100:6.2f}] + epilogue {r[6] / n / 100:5.2f}; tiles/block {r[7] / n:5.1f}")
