"""Generates tests/golden/*.npz by running the REFERENCE implementation (via oracle/ref_shim.py) in the build container.

TEST INFRASTRUCTURE.  Run:  python -m oracle.gen_golden   (needs /root/reference; CPU only, ~1 min)

Each case = a tiny-width VSM with the real topology, seeded synthetic weights (vstar_amd.weights.random_state_dict —
regenerated bit-identically from the seed by the tests, so the fixture stores only inputs' seeds and the reference's
outputs), and one or more crops each pushed through the reference's own `model_forward(inference=True)`
(VisualSearch/model/VSM.py:201-364) with batch 1, exactly as `visual_search.py` drives it.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import ref_shim  # noqa: E402
from vstar_amd.config import VSMConfig  # noqa: E402
from vstar_amd.weights import random_state_dict  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

CASES = {
    # name: (cfg kwargs, weight seed, [(input seed, L, image col, loc col)])
    "tiny224": (dict(), 0, [(1, 24, 5, 21), (2, 24, 5, 21), (3, 24, 5, 20)]),
    "tiny224_long": (dict(), 7, [(11, 31, 9, 28)]),
    "tiny336": (dict(clip_image_size=336), 3, [(21, 20, 4, 17), (22, 20, 4, 17)]),
}


def make_inputs(cfg: VSMConfig, seed: int, L: int, img_col: int, loc_col: int, loc_id: int):
    g = torch.Generator().manual_seed(seed)
    I = cfg.clip_image_size
    clip = torch.randn(1, 3, I, I, generator=g)
    owl = torch.randn(1, 3, cfg.owl_image_size, cfg.owl_image_size, generator=g)
    ids = torch.randint(3, loc_id - 3, (1, L), generator=g)
    ids[0, 0] = 1
    ids[0, img_col] = -200
    ids[0, loc_col] = loc_id
    return clip, owl, ids


def main():
    assert ref_shim.available(), "reference tree not found"
    os.makedirs(OUT, exist_ok=True)
    for name, (kw, wseed, crops) in CASES.items():
        cfg = VSMConfig.tiny(**kw)
        loc_id = cfg.llm_vocab - 1
        rec = {k: [] for k in ("pred_logits", "pred_boxes", "low_res_masks", "clip_features", "llm_hidden_loc",
                               "embed_det", "embed_seg", "ids", "in_checksum", "sam_hyper", "sam_upscaled_mean")}
        P = cfg.n_img_tokens
        for (seed, L, img_col, loc_col) in crops:
            # A fresh reference model per crop: under transformers 5.x the reference's nested HF forwards leave
            # output-capture state behind, and a SECOND model_forward on the same module object returns
            # hidden_states[-1] instead of [-2] from the CLIP tower (observed; a harness artefact of running the
            # 4.31-era reference on 5.x, not reference semantics).  First calls are exact (they match the oracle to 1e-6).
            _, model = ref_shim.load_reference(cfg, loc_id)
            sd = random_state_dict(cfg, seed=wseed, dtype=torch.float32)
            missing = ref_shim.load_state(model, sd)
            assert not missing, missing
            taps = {}
            vt = model.get_model().get_vision_tower()
            def hook_clip(m, i, o):
                taps["clip_features"] = o.detach().clone()

            def hook_det(m, i, o):
                taps["hidden"] = i[0].detach().clone()
                taps["det_all"] = o.detach().clone()

            def hook_seg(m, i, o):
                taps["seg_all"] = o.detach().clone()

            h1 = vt.register_forward_hook(hook_clip)
            h2 = model.model.text_hidden_fcs_det[0].register_forward_hook(hook_det)
            h3 = model.model.text_hidden_fcs_seg[0].register_forward_hook(hook_seg)
            # operands of the final mask product (mask_decoder.py:176-186): hyper_in [32] and the upscaled embedding [32,192,192]
            # (only its per-channel mean is stored: the common-mode vector that conditions the mask's offset)
            md = model.model.mask_decoder
            md.output_hypernetworks_mlps[0].register_forward_hook(lambda m, i, o: taps.__setitem__("sam_hyper", o.detach().float().clone()))
            md.output_upscaling.register_forward_hook(lambda m, i, o: taps.__setitem__("sam_up", o.detach().float().clone()))
            clip, owl, ids = make_inputs(cfg, seed, L, img_col, loc_col, loc_id)
            out = ref_shim.reference_forward(model, clip, owl, ids)
            pos = loc_col - 1 + (P - 1)
            rec["pred_logits"].append(out["pred_logits"][0, :, 0].numpy())
            rec["pred_boxes"].append(out["pred_boxes"][0].numpy())
            rec["low_res_masks"].append(out["pred_masks"][0][0].numpy())
            rec["clip_features"].append(taps["clip_features"][0].numpy())
            rec["llm_hidden_loc"].append(taps["hidden"][0, pos].numpy())
            rec["embed_det"].append(taps["det_all"][0, pos].numpy())
            rec["embed_seg"].append(taps["seg_all"][0, pos].numpy())
            rec["sam_hyper"].append(taps["sam_hyper"].reshape(-1).numpy())
            rec["sam_upscaled_mean"].append(taps["sam_up"][0].double().mean(dim=(1, 2)).float().numpy())
            rec["ids"].append(ids[0].numpy().astype(np.int32))
            rec["in_checksum"].append(np.array([clip.double().sum().item(), owl.double().sum().item()]))
        meta = dict(weight_seed=wseed, crops=np.array(crops, dtype=np.int64), loc_id=loc_id,
                    cfg_keys=np.array(sorted(kw.keys())), cfg_vals=np.array([kw[k] for k in sorted(kw.keys())], dtype=np.int64))
        arrays = {k: np.stack(v).astype(np.float32 if k not in ("ids",) else np.int32) if k != "in_checksum" else np.stack(v)
                  for k, v in rec.items()}
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **arrays, **meta)
        print(name, {k: v.shape for k, v in arrays.items()}, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
