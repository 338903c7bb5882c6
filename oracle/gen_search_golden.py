"""Records the REFERENCE scheduler's decisions (visual_search.py:484-516 run as-is) on synthetic images with FakeVSM.
TEST INFRASTRUCTURE.  Run in the build container:  python -m oracle.gen_search_golden"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.search_oracle import FakeVSM, load_reference_search  # noqa: E402
from vstar_amd.synthetic import synthetic_image  # noqa: E402,F401  (the generator lives in the product package; re-exported)

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "search_paths.json")

# (W, H, image seed, vsm seed, conf_shift, minimum_size_scale)
CASES = [(1920, 1080, 0, 0, -2.5, 4.0), (1920, 1080, 1, 1, -6.0, 4.0), (3840, 2160, 2, 2, -6.0, 4.0), (900, 2400, 3, 3, -4.0, 4.0),
         (2400, 700, 4, 4, -4.0, 4.0), (1500, 1500, 5, 5, -1.0, 4.0), (2048, 1536, 6, 6, -6.0, 8.0), (640, 480, 7, 7, -6.0, 4.0),
         (1920, 1080, 8, 8, 2.0, 4.0)]


# near-tie cases: mirror-symmetric heat maps (FakeVSM(symmetric=True)); even widths so that the 2x2 split is exactly mirrored
SYM_CASES = [(1920, 1080, 11, 11, -6.0, 4.0), (2048, 1536, 12, 12, -6.0, 8.0), (3840, 2160, 13, 13, -6.0, 4.0)]

CUE_TEXT = "The object is most likely to appear on the wooden table near the window."


def run_case(search_fn, case, cue=False, symmetric=False):
    w, h, iseed, vseed, shift, scale = case
    img = synthetic_image(w, h, iseed)
    smallest = max(int(np.ceil(min(w, h) / scale)), 224)
    # cue=True: detection heatmaps stay below the cue threshold, so every expanded node takes the contextual-cue branch
    # (VQA text -> phrase -> segmentation heatmap); the reference's spaCy is stubbed to return no tokens, so its
    # noun-chunk list is empty and the phrase becomes "region <phrase>" (visual_search.py:437-440)
    vsm = FakeVSM(seed=vseed, conf_shift=shift, gain=0.1 if cue else 9.0, vqa_text=CUE_TEXT if cue else None, symmetric=symmetric)
    # gain 9 keeps heat.max() above the decayed cue threshold so the (unbuilt) contextual-cue branch is not taken
    final_step, path_length, ok, all_valid = search_fn(vsm, img, "object", [0, 0, 10, 10], smallest)
    if cue:
        extra = {"cue": True, "n_vqa": sum(1 for m, _ in vsm.questions if m == "vqa"),
                 "seg_questions": sorted({q for m, q in vsm.questions if m == "segmentation"})}
    else:
        extra = {"cue": False, "symmetric": True} if symmetric else {"cue": False}
    return {**extra, "case": list(case), "smallest_size": smallest, "calls": vsm.calls, "path_length": int(path_length), "success": bool(ok),
            "final_bbox": [int(v) for v in final_step["bbox"]],
            "detection_result": [float(v) for v in final_step["detection_result"]],
            "n_all_valid": None if all_valid is None else int(all_valid.shape[0])}


def main():
    ref = load_reference_search()
    out = [run_case(ref.visual_search, c) for c in CASES]
    out += [run_case(ref.visual_search, c, cue=True) for c in CASES[1:3]]
    out += [run_case(ref.visual_search, c, symmetric=True) for c in SYM_CASES]
    json.dump(out, open(OUT, "w"), indent=1)
    for o in out:
        print(o)


if __name__ == "__main__":
    main()
