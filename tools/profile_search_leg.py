#!/usr/bin/env python
"""cProfile of bench.py's search leg (host-side hot spots).  usage: python tools/profile_search_leg.py [grouped|plain]"""
import cProfile, io, os, pstats, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from vstar_amd.config import VSMConfig
from vstar_amd.engine import VstarEngine
from vstar_amd.weights import random_state_dict
group = (sys.argv[1] if len(sys.argv) > 1 else "grouped") == "grouped"
cfg = VSMConfig.seal_7b(336, max_batch=32, max_text_len=65)
eng = VstarEngine(cfg, 0)
eng.load_state_dict(random_state_dict(cfg, seed=0, dtype=torch.bfloat16, share_layers=True))
args = types.SimpleNamespace(config5=False, search_targets=16, rccl_selfcheck=False)
bench.search_leg(eng, cfg, args, 0, group=group)          # warm
pr = cProfile.Profile()
pr.enable()
out = bench.search_leg(eng, cfg, args, 0, group=group)
pr.disable()
print({k: out[k] for k in ("search_crops_per_s", "wall_s", "stage_s")})
st = io.StringIO()
pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(22)
print("\n".join(l for l in st.getvalue().splitlines() if l.strip())[:6000])
