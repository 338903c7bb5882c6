"""Generates tests/golden/vqa_*.npz by running the REFERENCE VQA-LLM (via oracle/vqa_ref_shim.py) in the build container.

TEST INFRASTRUCTURE.  Run:  python -m oracle.gen_vqa_golden   (needs /root/reference; CPU only, ~1 min)

Each case = a tiny-width LlavaSearchLlamaForCausalLM with the real topology (CLIP@224 head dim 64, LLaMA head dim 128,
Perceiver heads of 96), seeded synthetic weights (vstar_amd.weights.random_state_dict — regenerated bit-identically from
the seed by the tests; the fixture stores seeds, token ids and the reference's outputs), driven exactly like
vstar_bench_eval.py drives the model: encode_images / project_features, the question forward with use_cache=True, each
option forwarded against the question's past_key_values (:148-151) and scored with CrossEntropyLoss (:153-159), and a
greedy decode (the manual past_key_values loop generate() performs; temperature 0).
"""
from __future__ import annotations

import copy
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import vqa_ref_shim  # noqa: E402
from vstar_amd.config import VQAConfig  # noqa: E402
from vstar_amd.weights import random_state_dict  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

CASES = {
    # name: (cfg kwargs, weight seed, input seed, n_objects, images_long, objects_long, question length, option lengths)
    "vqa_tiny_objects": (dict(), 0, 1, 2, [False], [True, False], 22, [4, 1, 6]),
    "vqa_tiny_plain": (dict(), 5, 2, 0, None, None, 17, [3, 5]),
    "vqa_tiny_mlp2x": (dict(projector_type=1), 9, 3, 1, [True], [False], 12, [2, 2]),
}
N_GEN = 6


def make_inputs(cfg: VQAConfig, seed: int, n_obj: int, qlen: int, opt_lens):
    """Pixels are fp16-representable so that the engine's fp16 input is exact; ids avoid the special range."""
    g = torch.Generator().manual_seed(seed)
    I = cfg.clip_image_size
    pix = torch.randn(1 + n_obj, 3, I, I, generator=g).half().float()
    ids = torch.randint(3, cfg.llm_vocab - 1, (qlen,), generator=g).tolist()
    ids[0] = 1
    ids[3] = -200
    for j in range(n_obj):
        ids[8 + 4 * j] = -300
    opts = [torch.randint(3, cfg.llm_vocab - 1, (n,), generator=g).tolist() for n in opt_lens]
    return pix, ids, opts


def main():
    assert vqa_ref_shim.available(), "reference tree not found"
    os.makedirs(OUT, exist_ok=True)
    for name, (kw, wseed, iseed, n_obj, images_long, objects_long, qlen, opt_lens) in CASES.items():
        cfg = VQAConfig.tiny(**kw)
        sd = random_state_dict(cfg, wseed, torch.float32)
        _, model = vqa_ref_shim.load_reference(cfg)
        assert not vqa_ref_shim.load_state(model, sd)
        pix, ids, opts = make_inputs(cfg, iseed, n_obj, qlen, opt_lens)
        image, objs = pix[:1], (pix[1:] if n_obj else None)
        with torch.no_grad():
            il, ish = model.encode_images(image)                                            # llava_search_arch.py:84-88
            ol, osh = model.project_features(objs) if n_obj else (torch.zeros(0), torch.zeros(0))
            q = model(torch.tensor([ids]), use_cache=True, images=image, object_features=objs, images_long=images_long,
                      objects_long=objects_long)                                            # vstar_bench_eval.py:127-133
            q_logits = q.logits[0]
            losses, opt_logits = [], []
            for o in opts:
                # transformers 4.31 (the reference's pin) hands out immutable tuple caches, so every option is scored
                # against the QUESTION's cache; 5.x DynamicCache objects are extended in place by a forward, so a copy
                # per option restores the reference semantics in this harness
                oo = model(input_ids=torch.tensor([o]), use_cache=True,
                           attention_mask=torch.ones(1, q_logits.shape[0] + len(o)),
                           past_key_values=copy.deepcopy(q.past_key_values))
                lg = torch.cat([q.logits[:, -1:], oo.logits[:, :-1]], 1)                    # :153
                losses.append(torch.nn.CrossEntropyLoss()(lg.view(-1, cfg.llm_vocab), torch.tensor(o)))   # :155-159
                opt_logits.append(oo.logits[0])
            # greedy decode with the KV cache (what generate(do_sample=False, use_cache=True) does step by step)
            q2 = model(torch.tensor([ids]), use_cache=True, images=image, object_features=objs, images_long=images_long,
                       objects_long=objects_long)
            past, last = q2.past_key_values, q2.logits[0, -1]
            gen, margins = [], []
            for _ in range(N_GEN):
                top = last.topk(2)
                gen.append(int(top.indices[0]))
                margins.append(float(top.values[0] - top.values[1]))
                step = model(input_ids=torch.tensor([[gen[-1]]]), use_cache=True,
                             attention_mask=torch.ones(1, q_logits.shape[0] + len(gen)), past_key_values=past)
                past, last = step.past_key_values, step.logits[0, -1]
        np.savez_compressed(
            os.path.join(OUT, name + ".npz"),
            cfg_kw=np.array(repr(kw)), weight_seed=wseed, input_seed=iseed, n_obj=n_obj,
            images_long=np.array([-1] if images_long is None else [int(b) for b in images_long]),
            objects_long=np.array([-1] if objects_long is None else [int(b) for b in objects_long]),
            ids=np.array(ids, np.int32), opt_lens=np.array(opt_lens, np.int32), opts=np.concatenate(opts).astype(np.int32),
            in_checksum=np.array([float(pix.double().sum()), float(pix.double().abs().sum())]),
            image_long=il[0].numpy().astype(np.float16), image_short=ish[0].numpy().astype(np.float16),
            obj_long=ol.numpy().astype(np.float16), obj_short=osh.numpy().astype(np.float16),
            q_logits_last=q_logits[-1].numpy(), q_logits_rows=q_logits[::16].numpy().astype(np.float16),
            opt_logits=torch.cat(opt_logits, 0).numpy().astype(np.float16),
            losses=torch.stack(losses).numpy(), gen=np.array(gen, np.int32), gen_margin=np.array(margins, np.float32))
        print(name, "S =", q_logits.shape[0], "losses", [round(float(x), 4) for x in losses], "gen", gen,
              "min margin %.4f" % min(margins))


if __name__ == "__main__":
    main()
