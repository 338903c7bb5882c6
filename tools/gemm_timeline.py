#!/usr/bin/env python
"""Where a gemm256 tile's time goes (diagnostic build -DGEMM_TIMELINE: tools/build_variant.sh tl "-DGEMM_TIMELINE" gemm256):
wall-clock stamps (s_memrealtime, 100 MHz) of wave 0 and wave 4 of workgroups 0 and 100 at
  0 tile start | 1 prologue done (first K-tile landed, barriers passed) | 2 K loop done | 3 trailing barrier | 4 next tile's first
  K-tile issued | 5 epilogue done
usage: VSTAR_LIB=vstar_amd/csrc/build/ab/lib_tl.so python tools/gemm_timeline.py [VSTAR_GEMM_DIRECT=0 for the LDS epilogue]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vstar_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
shapes = [("llama o +res", 20480, 4096, 4096, 0, 1, 0), ("llama o", 20480, 4096, 4096, 0, 0, 0), ("llama gate|up", 20480, 22016, 4096, 4, 0, 0),
          ("clip fc1", 18464, 4096, 1024, 1, 0, 1), ("owl out +res", 73760, 768, 768, 0, 1, 1), ("owl fc1", 73760, 3072, 768, 1, 0, 1)]
for name, M, N, K, epi, has_res, has_bias in shapes:
    a = torch.randn(M, K, device=dev).bfloat16()
    w = torch.zeros((N + 255) // 256 * 256, K, device=dev, dtype=torch.bfloat16)
    w[:N] = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    n_out = N // 2 if epi == 4 else N
    c = torch.empty(M, n_out, device=dev, dtype=torch.bfloat16)
    res = torch.randn(M, n_out, device=dev).bfloat16() if has_res else None
    bias = torch.randn(w.shape[0], device=dev).bfloat16() if has_bias else None
    for _ in range(3):
        assert lib.vstar_op_gemm(None, P(a), K, P(w), P(bias), P(res), n_out, P(c), n_out, 0, M, N, K, epi | 0x400) == 0
    torch.cuda.synchronize()
    buf = np.zeros((2, 2, 64, 8), np.uint64)
    assert lib.vstar_debug_gemm_timeline(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    t = buf.astype(np.int64)
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    per_cu = (tiles + 255) // 256
    print(f"\n{name}: M {M} N {N} K {K}: {tiles} tiles, {per_cu} per CU (10-ns ticks -> us)")
    for b in range(2):
        for g in range(2):
            rows = []
            for i in range(1, min(per_cu - 1, 6)):          # steady-state tiles (skip the first and the last)
                s = t[b, g, i]
                if s[0] == 0 or s[5] == 0:
                    continue
                nxt = t[b, g, i + 1][0]
                rows.append([(s[1] - s[0]) / 100, (s[2] - s[1]) / 100, (s[3] - s[2]) / 100, (s[4] - s[3]) / 100, (s[5] - s[4]) / 100,
                             (nxt - s[5]) / 100 if nxt else 0, (nxt - s[0]) / 100 if nxt else 0])
            if rows:
                r = np.mean(np.asarray(rows), axis=0)
                print(f"  wg {0 if b == 0 else 100} wave {4 * g}: prologue {r[0]:6.2f} | K loop {r[1]:7.2f} | trail bar {r[2]:5.2f} | next issue {r[3]:5.2f} | "
                      f"epilogue {r[4]:6.2f} | to next {r[5]:5.2f} || tile {r[6]:7.2f} us  ({len(rows)} tiles)")
