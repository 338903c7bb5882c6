"""Search-path visualisation of `visual_search(..., visualize=True, save_path=...)` (reference: visual_search.py:289-376).

The reference renders with cv2 (rectangles, Hershey text, COLORMAP_JET) and matplotlib; neither exists in this image, so the same
FILE SET is written with PIL + numpy:

    whole_image.jpg, step_{k}.jpg (the visited patch outlined in blue on the whole image, ground truth in red, labels as in the
    reference), step_{k}_heatmap.jpg (min-max normalised final heat map, JET colours, blended 0.5 / 0.5 with the patch and
    renormalised to its maximum like `show_heatmap_on_image`), final_patch_image.jpg, search_result.jpg (detection box on the final
    patch), context_cue.txt ('step{k}: <answer>#<phrase>' lines).

Rectangle geometry, colours, step numbering, the `search_length` cut-off and the heat-map arithmetic restate the reference; glyph
shapes are PIL's default bitmap font instead of Hershey Simplex (the one thing that cannot match pixel for pixel).  Host-side and
out of the hot path: nothing here touches the engine."""
from __future__ import annotations

import os
from typing import Sequence

import numpy as np
from PIL import Image, ImageDraw

BOX_COLOR = (255, 0, 0)
TEXT_COLOR = (255, 255, 255)


def _jet(v: np.ndarray) -> np.ndarray:
    """OpenCV's COLORMAP_JET as RGB floats in [0, 1] for v in [0, 1] (piecewise-linear ramps: blue -> cyan -> yellow -> red)."""
    v = np.clip(v, 0.0, 1.0)
    r = np.clip(1.5 - np.abs(4.0 * v - 3.0), 0.0, 1.0)
    g = np.clip(1.5 - np.abs(4.0 * v - 2.0), 0.0, 1.0)
    b = np.clip(1.5 - np.abs(4.0 * v - 1.0), 0.0, 1.0)
    return np.stack([r, g, b], axis=-1)


def visualize_bbox(img: np.ndarray, bbox: Sequence[float], class_name: str, color=BOX_COLOR, thickness: int = 2) -> np.ndarray:
    """One labelled box on an RGB uint8 array ([x, y, w, h]; visual_search.py:289-306): outline, filled label bar above the
    top-left corner, white text."""
    x_min, y_min, w, h = [float(v) for v in bbox]
    x_min, x_max, y_min, y_max = int(x_min), int(x_min + w), int(y_min), int(y_min + h)
    im = Image.fromarray(np.ascontiguousarray(img))
    d = ImageDraw.Draw(im)
    d.rectangle([x_min, y_min, x_max, y_max], outline=tuple(color), width=thickness)
    l, t, r, b = d.textbbox((0, 0), class_name)
    tw, th = r - l, b - t
    d.rectangle([x_min, y_min - int(1.3 * th) - 2, x_min + tw, y_min], fill=tuple(color))
    d.text((x_min, y_min - int(1.3 * th) - 2), class_name, fill=TEXT_COLOR)
    return np.asarray(im)


def vis_heatmap(image: np.ndarray, heatmap: np.ndarray) -> np.ndarray:
    """Min-max normalised heat map over the patch (visual_search.py:308-337, use_rgb=True, image_weight 0.5)."""
    hm = np.asarray(heatmap, np.float32)
    hm = hm.reshape(hm.shape[0], hm.shape[1])
    mx, mn = float(hm.max()), float(hm.min())
    if mx != mn:
        hm = (hm - mn) / (mx - mn)
    colours = _jet(np.uint8(255 * np.clip(hm, 0, 1)).astype(np.float32) / 255.0).astype(np.float32)
    cam = 0.5 * colours + 0.5 * (image.astype(np.float32) / 255.0)
    cam = cam / max(float(cam.max()), 1e-12)
    return np.uint8(255 * cam)


def visualize_search_path(image: Image.Image, search_path, search_length: int, target_bbox, label: str, save_path: str) -> None:
    """visual_search.py:339-376: one directory per (image, target)."""
    os.makedirs(save_path, exist_ok=True)
    image.save(os.path.join(save_path, "whole_image.jpg"))
    whole = np.array(image)
    if target_bbox is not None:
        whole = visualize_bbox(whole.copy(), target_bbox, class_name="gt: " + label, color=(255, 0, 0))
    cues = []
    for step_i, node in enumerate(search_path):
        if step_i + 1 > search_length:
            break
        x, y, w, h = node["bbox"]
        patch = image.crop((x, y, x + w, y + h))
        if "detection_result" in node:
            patch.save(os.path.join(save_path, "final_patch_image.jpg"))
            det = [float(v) for v in np.asarray(node["detection_result"], dtype=np.float64).reshape(-1)[:4]]
            Image.fromarray(visualize_bbox(np.array(patch), det, class_name="search result", color=(255, 0, 0))).save(
                os.path.join(save_path, "search_result.jpg"))
        cur = visualize_bbox(whole.copy(), node["bbox"], class_name="step-{}".format(step_i + 1), color=(0, 0, 255))
        Image.fromarray(cur).save(os.path.join(save_path, "step_{}.jpg".format(step_i + 1)))
        if "context_cue" in node:
            cues.append("step{}: {}".format(step_i + 1, node["context_cue"]) + "\n")
        if "final_heatmap" in node:
            Image.fromarray(vis_heatmap(np.array(patch), node["final_heatmap"])).save(
                os.path.join(save_path, "step_{}_heatmap.jpg".format(step_i + 1)))
    with open(os.path.join(save_path, "context_cue.txt"), "w") as f:
        f.writelines(cues)
