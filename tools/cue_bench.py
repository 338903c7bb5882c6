"""Contextual-cue text generation (VSM.inference mode='vqa', visual_search.py:427-443) at the 7B geometry: KV-cached decode
(vstar_vsm_generate) vs the reference's literal schedule (one full prefill per new token).  Prints one JSON object."""
import json
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vstar_amd import preprocess as pp  # noqa: E402
from vstar_amd.config import VSMConfig  # noqa: E402
from vstar_amd.vsm import VSM  # noqa: E402

n_new = int(os.environ.get("N_NEW", "100"))
cfg = VSMConfig.seal_7b(224, max_batch=1, max_text_len=256)
vsm = VSM(SimpleNamespace(version="synthetic", vision_tower="synthetic", conv_type="llava_v1", use_mm_start_end=True,
                          model_max_length=512), cfg=cfg, synthetic_seed=0)
vsm.vsm_tokenizer.eos_token_id = -1       # random weights: never stop early, time exactly n_new tokens
img = Image.fromarray(np.random.default_rng(0).integers(0, 256, (600, 800, 3), dtype=np.uint8))
q = pp.CUE_QUESTION.format("red umbrella")
out = {"new_tokens": n_new, "prompt_rows": None}
for name, uc, n in (("kv_cached", True, n_new), ("no_cache_reference_schedule", False, min(n_new, 20))):
    vsm.generate_ids(img, q, max_new_tokens=2, use_cache=uc)
    t0 = time.time()
    ids = vsm.generate_ids(img, q, max_new_tokens=n, use_cache=uc)
    dt = time.time() - t0
    assert len(ids) == n
    out[name] = {"tokens": n, "seconds": round(dt, 3), "ms_per_token": round(dt / n * 1e3, 2)}
out["speedup_per_token"] = round(out["no_cache_reference_schedule"]["ms_per_token"] / out["kv_cached"]["ms_per_token"], 2)
print(json.dumps(out))
