"""Records the prompts the REFERENCE builds (VisualSearch/model/llava/conversation.py templates, as driven by
visual_search.py:174-184) for both `--conv_type` values.  TEST INFRASTRUCTURE.  Run: python -m oracle.gen_prompt_golden
-> tests/golden/prompts.json (compared with vstar_amd.preprocess.build_prompt by tests/test_host.py)."""
from __future__ import annotations

import importlib.util
import json
import os

REF = os.environ.get("VSTAR_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "prompts.json")


def main():
    spec = importlib.util.spec_from_file_location("_ref_conversation", os.path.join(REF, "VisualSearch/model/llava/conversation.py"))
    conv_lib = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(conv_lib)
    out = []
    questions = ["Please locate the blue kite in this image.",
                 "According to the common sense knowledge and possible visual cues, what is the most likely location of the dog in the image?"]
    for conv_type in ("llava_v1", "llava_llama_2"):
        for use_mm in (True, False):
            for q in questions:
                for answer in ("", "Sure, [LOC]."):
                    conv = conv_lib.conv_templates[conv_type].copy()
                    conv.messages = []
                    prompt = "<image>" + "\n" + q
                    if use_mm:
                        prompt = prompt.replace("<image>", "<im_start><image><im_end>")
                    conv.append_message(conv.roles[0], prompt)
                    conv.append_message(conv.roles[1], answer)
                    out.append({"conv_type": conv_type, "use_mm_start_end": use_mm, "question": q, "answer": answer,
                                "prompt": conv.get_prompt()})
    json.dump(out, open(OUT, "w"), indent=1)
    print(len(out), "prompts ->", OUT)
    print(out[8]["prompt"])


if __name__ == "__main__":
    main()
