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


def bench_inputs(cfg, B: int, text_tokens: int, rank: int = 0):
    """The synthetic crop batch of bench.py (BASELINE config 2 shape), generated on the HOST so that the full-depth golden
    (oracle/gen_fulldepth_golden.py runs the reference on crops of exactly this batch) and the benchmark share one input set:
    clip [B,3,I,I] / owl [B,3,768,768] N(0,1) bf16, ids [B,L] int32 with BOS, one -200 and random vocabulary ids, the spliced
    [LOC]-1 position and three verify positions."""
    import torch
    L = text_tokens + 1
    P = cfg.n_img_tokens
    g = torch.Generator().manual_seed(1234 + rank)
    clip = torch.randn(B, 3, cfg.clip_image_size, cfg.clip_image_size, generator=g).bfloat16()
    owl = torch.randn(B, 3, cfg.owl_image_size, cfg.owl_image_size, generator=g).bfloat16()
    rng = np.random.default_rng(rank)
    ids = rng.integers(3, cfg.llm_vocab - 5, size=(B, L), dtype=np.int32)
    ids[:, 0] = 1
    ids[:, 35 if L > 40 else 2] = -200
    ids[:, L - 3] = cfg.llm_vocab - 1           # [LOC]: the last three columns stand for "[LOC] . </s>"
    loc = np.full((B,), (L - 3) - 1 + (P - 1), dtype=np.int32)
    verify = np.stack([loc, loc + 1, loc + 2], axis=1).astype(np.int32)
    return clip, owl, ids, loc, verify
