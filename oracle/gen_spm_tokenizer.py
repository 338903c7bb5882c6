"""A REAL sentencepiece/LLaMA tokenizer for the host-side tests (VERDICT r2 "missing #5"): there is no network, so no
`tokenizer.model` of craigwu/seal_vsm_7b — but `sentencepiece` is installed, so a tiny BPE model with byte fallback (the LLaMA
recipe: BOS=1, EOS=2, UNK=0, dummy prefix, identity normalisation) is trained here on a seeded synthetic corpus and wrapped
the way the checkpoint's tokenizer is: `[LOC]`, `<im_start>`, `<im_end>` added on top of the base vocabulary
(VisualSearch/model/VSM.py adds [LOC]; LLaVA adds the image markers), saved as a HF tokenizer directory.

TEST INFRASTRUCTURE.  Run in the build container:  python -m oracle.gen_spm_tokenizer
  -> tests/golden/spm_llama/{tokenizer.model, tokenizer.json, tokenizer_config.json}   (loaded by AutoTokenizer.from_pretrained
     exactly like visual_search.py:148-156 does: the `VSM.__init__` AutoTokenizer branch)
  -> tests/golden/spm_prompts.json: the ids the REFERENCE's own `tokenizer_image_token` (VisualSearch/model/llava/mm_utils.py:19-44)
     and conversation templates produce with this tokenizer for a set of questions — the pin for vstar_amd.preprocess and for
     VSM._ids (prompt / teacher-forced answer boundary, [LOC] id, template prefix).
"""
from __future__ import annotations

import importlib.util
import json
import os
import random
import shutil
import sys
import tempfile
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("VSTAR_REFERENCE", "/root/reference")
OUT_DIR = os.path.join(ROOT, "tests", "golden", "spm_llama")
OUT_JSON = os.path.join(ROOT, "tests", "golden", "spm_prompts.json")

OBJECTS = ["red cup", "zebra", "anchor", "theatre", "dog", "blue kite", "person in a yellow coat", "the", "orange traffic cone",
           "cup.", "e-scooter", "7up can", "naïve sign", "(small) box", "  two  spaces", "Étoile", "umbrella umbrella"]


def corpus(seed: int = 0):
    rng = random.Random(seed)
    words = ("a chat between curious user and an artificial intelligence assistant the gives helpful detailed polite answers to "
             "questions please locate in this image sure according common sense knowledge possible visual cues what is most likely "
             "location of red blue green yellow orange cup dog cat person bottle table chair left right region near on under behind "
             "kite umbrella coat traffic cone box sign can scooter window door tree car bus street shelf kitchen").split()
    lines = [" ".join(rng.choice(words) for _ in range(rng.randint(4, 20))) for _ in range(3000)]
    lines += ["A chat between a curious user and an artificial intelligence assistant. The assistant gives helpful, detailed, and "
              "polite answers to the user's questions. USER: Please locate the red cup in this image. ASSISTANT: Sure, it is here."] * 40
    lines += ["According to the common sense knowledge and possible visual cues, what is the most likely location of the dog in the image?"] * 20
    return lines


def build_tokenizer(out_dir: str):
    import sentencepiece as spm
    from transformers import AutoTokenizer
    tmp = tempfile.mkdtemp()
    try:
        with open(os.path.join(tmp, "corpus.txt"), "w") as f:
            f.write("\n".join(corpus()))
        spm.SentencePieceTrainer.train(input=os.path.join(tmp, "corpus.txt"), model_prefix=os.path.join(tmp, "tokenizer"),
                                       vocab_size=512, model_type="bpe", byte_fallback=True, character_coverage=1.0, unk_id=0,
                                       bos_id=1, eos_id=2, pad_id=-1, normalization_rule_name="identity", add_dummy_prefix=True,
                                       split_digits=True, minloglevel=2, num_threads=1)
        base = os.path.join(tmp, "base")
        os.makedirs(base)
        shutil.copy(os.path.join(tmp, "tokenizer.model"), os.path.join(base, "tokenizer.model"))
        json.dump({"tokenizer_class": "LlamaTokenizer", "add_bos_token": True, "add_eos_token": False, "bos_token": "<s>",
                   "eos_token": "</s>", "unk_token": "<unk>", "legacy": True}, open(os.path.join(base, "tokenizer_config.json"), "w"))
        tok = AutoTokenizer.from_pretrained(base, use_fast=False, model_max_length=512, padding_side="right")
        tok.add_tokens("[LOC]")                                             # VisualSearch/train: tokenizer.add_tokens("[LOC]")
        tok.add_tokens(["<im_start>", "<im_end>"], special_tokens=True)      # mm_use_im_start_end
        shutil.rmtree(out_dir, ignore_errors=True)
        tok.save_pretrained(out_dir)
        shutil.copy(os.path.join(tmp, "tokenizer.model"), os.path.join(out_dir, "tokenizer.model"))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return AutoTokenizer.from_pretrained(out_dir, use_fast=False, model_max_length=512, padding_side="right")


def reference_modules():
    """The reference's conversation templates and tokenizer_image_token, imported from where they lie."""
    spec = importlib.util.spec_from_file_location("_ref_conversation", os.path.join(REF, "VisualSearch/model/llava/conversation.py"))
    conv_lib = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(conv_lib)
    # mm_utils imports `.constants` relatively: give it a package
    pkg = types.ModuleType("_ref_llava")
    pkg.__path__ = [os.path.join(REF, "VisualSearch/model/llava")]
    sys.modules["_ref_llava"] = pkg
    mm = importlib.import_module("_ref_llava.mm_utils")
    return conv_lib, mm


def main():
    tok = build_tokenizer(OUT_DIR)
    tok.pad_token = tok.unk_token
    conv_lib, mm = reference_modules()
    loc = tok("[LOC]", add_special_tokens=False).input_ids
    assert len(loc) == 1, loc
    out = {"vocab": len(tok), "loc_token_idx": loc[0], "bos": tok.bos_token_id, "eos": tok.eos_token_id, "cases": []}
    questions = [f"Please locate the {o} in this image." for o in OBJECTS]
    questions += ["According to the common sense knowledge and possible visual cues, what is the most likely location of the dog in the image?",
                  "Please locate the region on the kitchen shelf in this image."]
    for conv_type in ("llava_v1", "llava_llama_2"):
        for q in questions:
            rec = {"conv_type": conv_type, "question": q}
            for key, answer in (("prompt_ids", ""), ("full_ids", "Sure, [LOC].")):
                conv = conv_lib.conv_templates[conv_type].copy()
                conv.messages = []
                conv.append_message(conv.roles[0], "<im_start><image><im_end>" + "\n" + q)
                conv.append_message(conv.roles[1], answer)
                prompt = conv.get_prompt()
                if answer:      # the teacher-forced sequence stops after the answer: drop the closing separator the template appends
                    sep2 = conv.sep2 if conv.sep2 else ""
                    stripped = prompt[: -len(sep2)] if sep2 and prompt.endswith(sep2) else prompt
                    prompt = stripped.rstrip(" ") if conv_type == "llava_llama_2" else stripped
                rec[key] = [int(t) for t in mm.tokenizer_image_token(prompt, tok)]
            out["cases"].append(rec)
    # what greedy decoding of the template looks like as ids (the answer the checkpoint was trained to emit, VisualSearch/utils/utils.py:18-20)
    out["answer_ids"] = [int(t) for t in tok(" Sure, [LOC].", add_special_tokens=False).input_ids]
    json.dump(out, open(OUT_JSON, "w"), indent=0)
    print(len(out["cases"]), "cases ->", OUT_JSON, "| tokenizer dir:", sorted(os.listdir(OUT_DIR)), "vocab", len(tok), "[LOC] =", loc[0])
    print(tok.convert_ids_to_tokens([t for t in out["cases"][1]["full_ids"] if t >= 0]))


if __name__ == "__main__":
    main()
