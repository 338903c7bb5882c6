#!/usr/bin/env python
"""VSTAR_F_SHARE_PREFIX at the 7B geometry: one 32-crop batch whose rows carry the same Lp tokens before <image> (the system
prompt of every real call), scored with and without the flag, interleaved.   usage: python tools/prefix_bench.py [Lp]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vstar_amd.config import VSMConfig
from vstar_amd.engine import VstarEngine
from vstar_amd.weights import random_state_dict
Lp = int(sys.argv[1]) if len(sys.argv) > 1 else 37
B, T = 32, 64
cfg = VSMConfig.seal_7b(336, max_batch=B, max_text_len=T + 1)
eng = VstarEngine(cfg, 0)
eng.load_state_dict(random_state_dict(cfg, seed=0, dtype=torch.bfloat16, share_layers=True))
g = torch.Generator().manual_seed(0)
clip = torch.randn(B, 3, 336, 336, generator=g).bfloat16().cuda()
owl = torch.randn(B, 3, 768, 768, generator=g).bfloat16().cuda()
L = T + 1
ids = np.random.default_rng(0).integers(3, 30000, (B, L)).astype(np.int32)
ids[:, :Lp] = ids[0, :Lp]; ids[:, 0] = 1; ids[:, Lp] = -200
loc = np.full(B, L - 4 + 575, np.int32)
res = {}
for rep in range(3):
    for share in (False, True):
        eng.score_batch(clip, owl, ids, loc, share_prefix=share, raw=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(4): eng.score_batch(clip, owl, ids, loc, share_prefix=share, raw=True)
        res.setdefault(share, []).append((time.perf_counter() - t0) / 4)
for share in (False, True):
    ms = min(res[share]) * 1e3
    print(f"share_prefix={share}: {ms:.1f} ms per 32-crop batch = {B / ms * 1e3:.1f} crops/s   (all: {[round(x * 1e3, 1) for x in res[share]]})")
