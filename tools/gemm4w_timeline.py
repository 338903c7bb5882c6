#!/usr/bin/env python
"""Where a gemm4w tile's time goes (diagnostic build: tools/build_variant.sh tl4 "-DG4W_TIMELINE" gemm4w; the no-store arm of
profiles/r06_gemm4w_timeline.txt was a switch of the epilogue at that commit):
wall-clock stamps (s_memrealtime, 100 MHz) of wave 0 of workgroups 0 and 100 at
  0 tile start (before the loop statement) | 1 K loop done | 2 next tile's pipeline head issued | 3 epilogue issued (stores in flight)
  | 4 everything this wave issued has completed (vmcnt(0))
usage: VSTAR_LIB=vstar_amd/csrc/build/ab/lib_tl4.so python tools/gemm4w_timeline.py"""
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
shapes = [("llama qkv", 20480, 12288, 4096, 0, 0), ("llama o +res", 20480, 4096, 4096, 0, 1), ("llama gate|up silu", 20480, 22016, 4096, 4, 0),
          ("llama down +res", 20480, 4096, 11008, 0, 1)]
for name, M, N, K, epi, has_res in shapes:
    a = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    n_out = N // 2 if epi == 4 else N
    c = torch.empty(M, n_out, device=dev, dtype=torch.bfloat16)
    res = torch.randn(M, n_out, device=dev).bfloat16() if has_res else None
    for _ in range(3):
        assert lib.vstar_op_gemm(None, P(a), K, P(w), None, P(res), n_out, P(c), n_out, 0, M, N, K, epi | _lib.EPI_TILE4W) == 0
    torch.cuda.synchronize()
    buf = np.zeros((2, 64, 8), np.uint64)
    assert lib.vstar_debug_gemm4w_timeline(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    t = buf.astype(np.int64)
    per_cu = (M // 256) * (N // 256) // 256
    print(f"\n{name}: M {M} N {N} K {K}: {per_cu} tiles per CU (us)")
    for b in range(2):
        rows = []
        for i in range(1, min(per_cu - 1, 12)):
            s, nxt = t[b, i], t[b, i + 1]
            if s[0] == 0 or nxt[0] == 0:
                continue
            rows.append([(s[1] - s[0]) / 100, (s[2] - s[1]) / 100, (s[3] - s[2]) / 100, (s[4] - s[3]) / 100, (nxt[0] - s[4]) / 100, (nxt[0] - s[0]) / 100])
        if rows:
            r = np.mean(np.asarray(rows), axis=0)
            print(f"  wg {0 if b == 0 else 100}: loop statement (zeroing, wait, K loop) {r[0]:7.2f} | head issue {r[1]:5.2f} | epilogue issue {r[2]:5.2f} | "
                  f"drain (vmcnt 0) {r[3]:5.2f} | set-up {r[4]:5.2f} || tile {r[5]:7.2f}  ({len(rows)} tiles)")
