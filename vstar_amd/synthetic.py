"""Synthetic inputs for benchmarks and fixtures (there are no datasets or checkpoints in the build/bench environment).

`synthetic_image` is the generator behind the scheduler goldens (oracle/gen_search_golden.py records the reference's search
paths on these images) and behind the search leg of bench.py: smooth colour blobs (a low-resolution random grid, bilinearly
enlarged), so crops of different regions differ and resizing is well conditioned."""
from __future__ import annotations

import numpy as np
from PIL import Image


def synthetic_image(w: int, h: int, seed: int) -> Image.Image:
    rng = np.random.default_rng(seed)
    low = rng.integers(0, 255, size=(h // 64 + 1, w // 64 + 1, 3), dtype=np.uint8)
    return Image.fromarray(low).resize((w, h), Image.BILINEAR)
