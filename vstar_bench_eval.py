#!/usr/bin/env python
"""V*Bench end-to-end entry point (reference: vstar_bench_eval.py:168-293), both models on the HIP engines:
the SEAL VQA-LLM (`vstar_amd.vqa.VQA_LLM`, fp16, KV-cached) and the visual-search model (`vstar_amd.vsm.VSM`, bf16).
`--vqa-model-path` / `--vsm-model-path` are local HF checkpoint directories (there is no hub access in this environment);
`--vqa-llm module:factory` substitutes another object with the reference's VQA_LLM interface.
"""
from __future__ import annotations

import argparse
import importlib
import sys


def parse_args(argv):
    p = argparse.ArgumentParser()
    p.add_argument("--vqa-model-path", type=str, default="craigwu/seal_vqa_7b")
    p.add_argument("--vqa-model-base", type=str, default=None)
    p.add_argument("--conv_type", default="v1", type=str)
    p.add_argument("--benchmark-folder", type=str, default="vstar_bench")
    p.add_argument("--vsm-model-path", type=str, default="craigwu/seal_vsm_7b")
    p.add_argument("--output-path", type=str, default="eval_result.json")
    p.add_argument("--minimum_size_scale", default=4.0, type=float)
    p.add_argument("--minimum_size", default=224, type=int)
    p.add_argument("--vision-tower", dest="vision_tower", default=None, help="local openai/clip-vit-large-patch14 directory")
    p.add_argument("--vqa-llm", default=None, help="module:factory providing another VQA-LLM implementation")
    p.add_argument("--vsm-factory", default=None, help="module:factory(args, device) providing another VSM implementation")
    p.add_argument("--device", default=0, type=int)
    p.add_argument("--search-window", dest="search_window", default=0, type=int, help="concurrent visual searches per engine batch "
                   "(cross-image lock step); 0 = one engine batch, 1 = one image at a time like the reference")
    p.add_argument("--engine-comm", nargs="?", const="on", default="auto", choices=["auto", "on", "off"],
                   help="world > 1 on GPUs: gather the per-step records with the C-ABI's own RCCL communicator on the engine stream "
                   "(vstar_allgather_results) instead of torch.distributed.  auto (default, round 6): used when the communicator comes up "
                   "and its self-check against torch.distributed passes on every rank, otherwise every rank falls back together; off = "
                   "torch.distributed")
    return p.parse_args(argv)


def main(argv):
    """Single process: `python vstar_bench_eval.py ...`.  One node, N GPUs (BASELINE configs 3/4):
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 vstar_bench_eval.py ...` — one process
    per GPU, weights replicated, each visual-search engine step crop-sharded with an RCCL all-gather of the records."""
    args = parse_args(argv)
    from vstar_amd.dist import finalize, init_from_env
    world, rank, local_rank = init_from_env()
    finished = False
    try:
        if world > 1:
            args.device = local_rank
        if args.vqa_llm:
            mod, fn = args.vqa_llm.split(":")
            vqa_llm = getattr(importlib.import_module(mod), fn)(args)
        else:
            from vstar_amd.vqa import VQA_LLM
            vqa_llm = VQA_LLM(args, device=local_rank if world > 1 else args.device)
        from vstar_amd.bench_eval import eval_model, make_vsm
        vsm = None
        if args.vsm_factory:
            mod, fn = args.vsm_factory.split(":")
            vsm = getattr(importlib.import_module(mod), fn)(args, local_rank)
        elif world > 1:
            vsm = make_vsm(args, local_rank)
        if args.engine_comm != "off" and vsm is not None and world > 1:
            from vstar_amd.dist import maybe_engine_comm
            maybe_engine_comm(vsm)
        eval_model(args, vqa_llm, vsm, world=world, rank=rank)
        finished = True
    finally:
        finalize(finished)


if __name__ == "__main__":
    main(sys.argv[1:])
