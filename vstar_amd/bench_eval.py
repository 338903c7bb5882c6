"""SEAL orchestration for V*Bench (reference: vstar_bench_eval.py:168-280): VQA-LLM free-form answer -> parse the
"missing objects" list -> visual search per object on the HIP engine -> object crops + focus prompt -> option ranking.
`vqa_llm` is any object with the reference's VQA_LLM interface (free_form_inference / multiple_choices_inference /
get_object_crop / image_processor); `vstar_amd.vqa.VQA_LLM` is that class on the HIP engine."""
from __future__ import annotations

import json
import os
from collections import defaultdict
from copy import deepcopy
from types import SimpleNamespace

import numpy as np
import torch
from PIL import Image

from .config import VSMConfig
from .search import smallest_size_for, visual_search, visual_search_many, visual_search_stream
from .vsm import VSM

MISSING_MSG = ("Sorry, I can not answer the question. Some visual information about the following objects is missing or "
               "unclear:")
FOCUS_MSG = "Additional visual information to focus on: "


def expand2square_centered(img: Image.Image, color):
    """The VQA-LLM side pads CENTRED and reports the offsets (vstar_bench_eval.py:25-36)."""
    w, h = img.size
    if w == h:
        return img, 0, 0
    side = max(w, h)
    out = Image.new(img.mode, (side, side), color)
    left, top = ((0, (w - h) // 2) if w > h else ((h - w) // 2, 0))
    out.paste(img, (left, top))
    return out, left, top


def parse_missing_objects(prediction: str):
    if MISSING_MSG not in prediction:
        return []
    tail = prediction.split(MISSING_MSG)[-1]
    if tail.endswith("."):
        tail = tail[:-1]
    return [t.strip() for t in tail.split(",")]


def search_objects(vsm, image_path, names, args):
    """vstar_bench_eval.py:205-209 (one visual_search per missing object); several missing objects of an image are searched in lock
    step (`visual_search_many`: same per-object tuples, the crops the objects share are scored together)."""
    found = []
    if not names:
        return found
    image = Image.open(image_path).convert("RGB")
    smallest = smallest_size_for(image.width, image.height, args.minimum_size_scale, args.minimum_size)
    if len(names) > 1:
        out = visual_search_many(vsm, image, names, None, smallest)
    else:
        out = [visual_search(vsm, image, names[0], target_bbox=None, smallest_size=smallest)]
    return _found(names, out)


def _found(names, out):
    """vstar_bench_eval.py:209-222: every valid box of a whole-image hit, else the final detection, in image coordinates."""
    found = []
    for name, (step, _, _, all_valid) in zip(names, out):
        boxes = all_valid if all_valid is not None else [step["detection_result"]]
        for b in boxes:
            b[0] += step["bbox"][0]
            b[1] += step["bbox"][1]
            found.append({"bbox": b.tolist(), "name": name})
    return found


def search_objects_stream(vsm, jobs, args, window=None, stats=None):
    """All (image, missing object) searches of an evaluation in ONE cross-image lock-step stream (visual_search_stream): jobs =
    [(image_path, [names])] -> one found-list per job, each equal to search_objects(vsm, path, names, args).  The reference runs
    them one after the other (vstar_bench_eval.py:190-262); here a window of searches shares every engine batch."""
    class _Loader:
        def __init__(self, path):
            self.key = path

        def __call__(self):
            return Image.open(self.key).convert("RGB")

    samples, owner = [], []
    for j, (path, names) in enumerate(jobs):
        ld = _Loader(path)
        for n in names:
            samples.append((ld, n, None, lambda im: smallest_size_for(im.width, im.height, args.minimum_size_scale, args.minimum_size)))
            owner.append(j)
    outs = visual_search_stream(vsm, samples, window=window, stats=stats)
    per = [[] for _ in jobs]
    for j, o in zip(owner, outs):
        per[j].append(o)
    return [_found(names, per[j]) for j, (_, names) in enumerate(jobs)]


def focus_question(question, found, image, pad_left, pad_top):
    parts = []
    for f in found:
        x, y, w, h = f["bbox"]
        x, y = x + pad_left, y + pad_top
        n = [float(np.clip(v, 0, 1)) for v in (x / image.width, y / image.height, (x + w) / image.width, (y + h) / image.height)]
        parts.append("{} <object> at location [{:.3f},{:.3f},{:.3f},{:.3f}]".format(f["name"], *n))
    return FOCUS_MSG + "; ".join(parts) + ".\n" + question


def make_vsm(args, device: int = 0):
    """The VSM of the evaluation loop (vstar_bench_eval.py:171-177).  `--vision-tower` must name a LOCAL
    openai/clip-vit-large-patch14 directory: the CLIP tower is not part of the VSM checkpoint and there is no hub access."""
    tower = getattr(args, "vision_tower", None) or "openai/clip-vit-large-patch14"
    if os.path.isdir(str(args.vsm_model_path)) and not os.path.isdir(str(tower)):
        raise FileNotFoundError(
            f"--vision-tower {tower!r} is not a local directory: stage openai/clip-vit-large-patch14 next to the VSM checkpoint and "
            "pass its path (the reference downloads it from the hub, visual_search.py:157-161; this environment has no network)")
    vsm_args = SimpleNamespace(version=args.vsm_model_path, vision_tower=tower, conv_type="llava_v1", use_mm_start_end=True,
                               model_max_length=512)
    return VSM(vsm_args, cfg=VSMConfig.seal_7b(224), device=device)


def eval_model(args, vqa_llm, vsm=None, world: int = 1, rank: int = 0):
    """world/rank: under torchrun (vstar_bench_eval.py) the VQA-LLM work is replicated on every rank (deterministic, so all
    ranks agree) and every visual-search engine step is crop-sharded over the ranks with the record all-gather;
    rank 0 prints and writes the results."""
    if vsm is None:
        vsm = make_vsm(args, int(getattr(args, "device", 0) or 0))       # (--device moved only the VQA-LLM)
    mean_color = tuple(int(x * 255) for x in vqa_llm.image_processor.image_mean)
    results, per_type, everything = {}, defaultdict(list), []
    # Three passes over independent questions instead of the reference's one pass (vstar_bench_eval.py:190-262): (1) the VQA-LLM's
    # free-form answers -> missing objects, (2) ALL visual searches in one cross-image lock-step stream (`--search-window`
    # concurrent searches per engine batch; 1 = one image at a time like the reference), (3) option ranking.  Per question the
    # values are the same; the engine sees full batches.
    window = int(getattr(args, "search_window", 0) or 0)
    max_found = getattr(args, "max_found_objects", None)        # (tests: tiny engines hold few object crops)
    entries = []
    for split in ("direct_attributes", "relative_position"):
        folder = os.path.join(args.benchmark_folder, split)
        for image_file in sorted(f for f in os.listdir(folder) if ".json" not in f):
            path = os.path.join(folder, image_file)
            ann = json.load(open(path.split(".")[0] + ".json"))
            square, _, _ = expand2square_centered(Image.open(path).convert("RGB"), mean_color)
            prediction = vqa_llm.free_form_inference(square, ann["question"])
            entries.append({"split": split, "image_file": image_file, "path": path, "question": ann["question"], "options": ann["options"],
                            "prediction": prediction, "missing": parse_missing_objects(prediction)})
    todo = [e for e in entries if e["missing"]]
    if window == 1:
        founds = [search_objects(vsm, e["path"], e["missing"], args) for e in todo]
    else:
        founds = search_objects_stream(vsm, [(e["path"], e["missing"]) for e in todo], args, window=window or None)
    for e, f in zip(todo, founds):
        e["found"] = f[:max_found] if max_found else f
    for split in ("direct_attributes", "relative_position"):
        results[split] = []
        for e in (x for x in entries if x["split"] == split):
            question, options, missing, found = e["question"], e["options"], e["missing"], e.get("found", [])
            image = Image.open(e["path"]).convert("RGB")
            if missing:
                crops = torch.stack([vqa_llm.get_object_crop(image, deepcopy(f["bbox"]), patch_scale=1.2) for f in found], 0)
                square, left, top = expand2square_centered(image, mean_color)
                long_objects = [len(found) <= 2] * len(found)
                chosen = vqa_llm.multiple_choices_inference(square, focus_question(question, found, square, left, top), options,
                                                            crops, images_long=[False], objects_long=long_objects)
            else:
                chosen = vqa_llm.multiple_choices_inference(image, question, options)
            correct = 1 if chosen == 0 else 0
            per_type[split].append(correct)
            everything.append(correct)
            results[split].append({"question": question, "options": options, "image": e["image_file"],
                                   "prediction_freeform": e["prediction"], "missing_objects": missing, "search_result": found,
                                   "option_chosen": chosen, "correct": correct})
        if rank == 0:
            print(split, np.mean(per_type[split]))
    if rank == 0:
        print(np.mean(everything))
        with open(args.output_path, "w") as f:
            json.dump(results, f, indent=4)
    return results
