#!/usr/bin/env python
"""Grouped scoring (vstar_vsm_score_grouped) vs the plain batch path at the 7B geometry: G crops x T prompts per call.
usage: python tools/grouped_bench.py [G T]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vstar_amd.config import VSMConfig
from vstar_amd.engine import VstarEngine
from vstar_amd.weights import random_state_dict
G, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2, 16)
cfg = VSMConfig.seal_7b(336, max_batch=32, max_text_len=65)
eng = VstarEngine(cfg, 0)
eng.load_state_dict(random_state_dict(cfg, seed=0, dtype=torch.bfloat16, share_layers=True))
g = torch.Generator().manual_seed(0)
clip = torch.randn(G, 3, 336, 336, generator=g).bfloat16().cuda()
owl = torch.randn(G, 3, 768, 768, generator=g).bfloat16().cuda()
Lp, Ls = 41, 24
prefix = np.random.default_rng(0).integers(3, 30000, Lp).astype(np.int32); prefix[0] = 1; prefix[35] = -200
suffix = np.random.default_rng(1).integers(3, 30000, (G, T, Ls)).astype(np.int32)
loc = np.full((G, T), Ls - 3, np.int32)
for _ in range(2): eng.score_grouped(clip, owl, prefix, suffix, loc, raw=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): eng.score_grouped(clip, owl, prefix, suffix, loc, raw=True)
dt = (time.perf_counter() - t0) / 5
print(f"grouped G={G} T={T}: {dt * 1e3:.1f} ms per call = {G * T / dt:.1f} records/s")
