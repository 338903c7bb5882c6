"""Generates tests/golden/*_bf16.npz: the REFERENCE implementation itself evaluated in bfloat16 (its production dtype,
visual_search.py:145 `torch_dtype=torch.bfloat16`) on the same seeded weights / inputs as the fp32 goldens of gen_golden.py.

TEST INFRASTRUCTURE.  Run:  python -m oracle.gen_golden_bf16   (needs /root/reference; CPU only, ~1 min)

Why: the engine computes in bf16 like the reference, so its distance from the fp32 goldens is dominated by bf16 rounding noise.
These vectors MEASURE that noise on the reference's own code (model.bfloat16(), bf16 inputs, torch-CPU kernels), so the GPU
parity gate can be stated against it:  err(engine, fp32 golden) <= 1.5 x err(reference-bf16, fp32 golden)  per tap
(tests/test_engine_gpu.py), with no fixed floor.  Harness note: transformers 5.x computes the CLIP eager softmax in
fp32 where the pinned 4.31 used the input dtype — this makes the recorded reference noise slightly SMALLER, i.e. the gate stricter.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import ref_shim  # noqa: E402
from oracle.gen_golden import CASES, OUT, make_inputs  # noqa: E402
from vstar_amd.config import VSMConfig  # noqa: E402
from vstar_amd.weights import random_state_dict  # noqa: E402

TAPS = ("pred_logits", "pred_boxes", "low_res_masks", "clip_features", "llm_hidden_loc", "embed_det", "embed_seg", "sam_hyper",
        "sam_upscaled_mean")


def reference_bf16_taps(cfg: VSMConfig, wseed: int, crop, loc_id: int):
    seed, L, img_col, loc_col = crop
    P = cfg.n_img_tokens
    _, model = ref_shim.load_reference(cfg, loc_id)          # fresh model per crop: see gen_golden.py
    missing = ref_shim.load_state(model, random_state_dict(cfg, seed=wseed, dtype=torch.float32))
    assert not missing, missing
    model = model.bfloat16()
    vt = model.get_model().get_vision_tower()
    vt.vision_tower = vt.vision_tower.bfloat16()
    taps = {}
    h1 = vt.register_forward_hook(lambda m, i, o: taps.__setitem__("clip_features", o.detach().float().clone()))

    def hook_det(m, i, o):
        taps["hidden"] = i[0].detach().float().clone()
        taps["det_all"] = o.detach().float().clone()
    h2 = model.model.text_hidden_fcs_det[0].register_forward_hook(hook_det)
    h3 = model.model.text_hidden_fcs_seg[0].register_forward_hook(lambda m, i, o: taps.__setitem__("seg_all", o.detach().float().clone()))
    md = model.model.mask_decoder
    h4 = md.output_hypernetworks_mlps[0].register_forward_hook(lambda m, i, o: taps.__setitem__("sam_hyper", o.detach().float().clone()))
    h5 = md.output_upscaling.register_forward_hook(lambda m, i, o: taps.__setitem__("sam_up", o.detach().float().clone()))
    clip, owl, ids = make_inputs(cfg, seed, L, img_col, loc_col, loc_id)
    out = ref_shim.reference_forward(model, clip.bfloat16(), owl.bfloat16(), ids)
    for h in (h1, h2, h3, h4, h5):
        h.remove()
    pos = loc_col - 1 + (P - 1)
    return {
        "pred_logits": out["pred_logits"][0, :, 0].float().numpy(),
        "pred_boxes": out["pred_boxes"][0].float().numpy(),
        "low_res_masks": out["pred_masks"][0][0].float().numpy(),
        "clip_features": taps["clip_features"][0].numpy(),
        "llm_hidden_loc": taps["hidden"][0, pos].numpy(),
        "embed_det": taps["det_all"][0, pos].numpy(),
        "embed_seg": taps["seg_all"][0, pos].numpy(),
        "sam_hyper": taps["sam_hyper"].reshape(-1).numpy(),
        "sam_upscaled_mean": taps["sam_up"][0].double().mean(dim=(1, 2)).float().numpy(),
    }


def main():
    assert ref_shim.available(), "reference tree not found"
    for name, (kw, wseed, crops) in CASES.items():
        cfg = VSMConfig.tiny(**kw)
        loc_id = cfg.llm_vocab - 1
        rec = {k: [] for k in TAPS}
        for crop in crops:
            t = reference_bf16_taps(cfg, wseed, crop, loc_id)
            for k in TAPS:
                rec[k].append(t[k])
        arrays = {k: np.stack(v).astype(np.float32) for k, v in rec.items()}
        fp32 = np.load(os.path.join(OUT, name + ".npz"))
        rel = lambda a, b: float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b.astype(np.float64)))  # noqa: E731
        print(name, {k: "%.2e" % max(rel(arrays[k][i], fp32[k][i]) for i in range(len(crops))) for k in TAPS})
        path = os.path.join(OUT, name + "_bf16.npz")
        np.savez_compressed(path, **arrays, weight_seed=wseed, crops=np.array(crops, dtype=np.int64))
        print("  ->", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
