#!/usr/bin/env python
"""V*Bench end-to-end entry point (reference: vstar_bench_eval.py:168-293).

The visual-search stage runs on the HIP engine.  The SEAL VQA-LLM (`load_pretrained_model`, `free_form_inference`,
`multiple_choices_inference`; LLaVA/llava/model/builder.py:26-151, vstar_bench_eval.py:38-165) is the next scope row
 and is NOT built yet: supply `--vqa-llm module:factory` returning an object with those two methods
(e.g. the reference's own VQA_LLM on another device) or this script stops with a clear error.
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
    p.add_argument("--vqa-llm", default=None, help="module:factory providing the VQA-LLM (not part of this engine yet)")
    return p.parse_args(argv)


def main(argv):
    args = parse_args(argv)
    if not args.vqa_llm:
        raise SystemExit("vstar_bench_eval: the SEAL VQA-LLM forward/generate is not built in this engine yet "
                         "(SURVEY.md §8f-2). Pass --vqa-llm module:factory, or run visual_search.py for the search stage.")
    mod, fn = args.vqa_llm.split(":")
    vqa_llm = getattr(importlib.import_module(mod), fn)(args)
    from vstar_amd.bench_eval import eval_model
    eval_model(args, vqa_llm)


if __name__ == "__main__":
    main(sys.argv[1:])
