"""Detection-mode fallback when greedy decoding does NOT emit "Sure, [LOC]." (SURVEY §7 "hard parts"; reference semantics:
VisualSearch/model/VSM.py:451-473 takes whatever generate() produced, visual_search.py:209-225 fails with IndexError when no
[LOC] came out).

The tiny model is CRAFTED so that its greedy decode is known: o_proj and down_proj are zero, so the residual stream at a
position is the embedding of that position's token, and lm_head rows are aligned with the embedding of the predecessor in a
chosen chain — a bigram model whose continuation of the prompt is exactly the chain."""
import warnings

import numpy as np
import pytest
import torch

from oracle import vsm_oracle
from vstar_amd import preprocess as pp
from vstar_amd.config import VSMConfig
from vstar_amd.engine import VstarEngine
from vstar_amd.synthetic import synthetic_image
from vstar_amd.vsm import VSM, DeferredMismatch
from vstar_amd.weights import random_state_dict

pytestmark = pytest.mark.gpu

CFG = VSMConfig.tiny(max_batch=4, max_text_len=192)
TOK = pp.SyntheticTokenizer(CFG.llm_vocab)
LOC, EOS = CFG.llm_vocab - 1, 2
QUESTION = pp.LOCATE_QUESTION.format("blue kite")


def _tid(piece):
    return TOK(piece, add_special_tokens=False).input_ids[0]


def bigram_state_dict(chain, seed=9):
    """chain: list of (token, next_token): after `token` the model's arg-max is `next_token`."""
    sd = random_state_dict(CFG, seed=seed, dtype=torch.bfloat16)
    for i in range(CFG.llm_layers):
        sd[f"model.layers.{i}.self_attn.o_proj.weight"].zero_()
        sd[f"model.layers.{i}.mlp.down_proj.weight"].zero_()
    sd["model.norm.weight"].fill_(1.0)
    E = sd["model.embed_tokens.weight"].float()
    g = torch.Generator().manual_seed(seed)
    W = 0.01 * torch.randn(CFG.llm_vocab, CFG.llm_hidden, generator=g)
    for t, nxt in chain:
        W[nxt] += E[t] / E[t].norm()
    sd["lm_head.weight"] = W.bfloat16()
    return sd


def make_vsm(chain):
    eng = VstarEngine(CFG, 0)
    sd = bigram_state_dict(chain)
    eng.load_state_dict(sd)
    return VSM(None, engine=eng, tokenizer=TOK, strict_template=True), sd


def _prompt_ids():
    return pp.tokenizer_image_token(pp.build_prompt(QUESTION), TOK)


def test_chain_tokens_are_distinct():
    ids = [_tid(p) for p in (":", "Sure", ",", ".", "Okay", "!", "here")] + [LOC, EOS]
    assert len(set(ids)) == len(ids), "SyntheticTokenizer hash collision: pick other chain words"


def test_template_emitted_no_fallback(cuda):
    colon, sure, comma, dot = _tid(":"), _tid("Sure"), _tid(","), _tid(".")
    vsm, _ = make_vsm([(colon, sure), (sure, comma), (comma, LOC), (LOC, dot), (dot, EOS)])
    img = synthetic_image(400, 300, 1)
    assert vsm.generate_ids(img, QUESTION, 20) == [sure, comma, LOC, dot, EOS]
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        boxes, scores, heat = vsm.inference(img, QUESTION, mode="detection")
    assert vsm.last_template_ok.all() and vsm.fallback_log == []
    assert boxes.shape == (2304, 4) and heat.shape == (300, 400)
    vsm.engine.close()


def test_other_wording_falls_back_to_stepwise_decode(cuda):
    """Greedy output "Okay [LOC] ! </s>": the template check fails, the engine decodes stepwise, finds the [LOC] the model
    really emitted and scores the crop with the GENERATED answer teacher-forced — identical to scoring those ids directly,
    and equal to the oracle on them."""
    colon, okay, bang = _tid(":"), _tid("Okay"), _tid("!")
    vsm, sd = make_vsm([(colon, okay), (okay, LOC), (LOC, bang), (bang, EOS)])
    img = synthetic_image(400, 300, 2)
    boxes, scores, heat = vsm.inference(img, QUESTION, mode="detection")
    assert not vsm.last_template_ok.any()
    assert len(vsm.fallback_log) == 1 and vsm.fallback_log[0]["generated"] == [okay, LOC, bang, EOS]
    assert vsm.fallback_log[0]["loc_positions"] == [1]
    # direct scoring of prompt + "Okay [LOC]"
    ids = np.asarray(_prompt_ids() + [okay, LOC], np.int32)
    P = CFG.n_img_tokens
    loc_pos = len(ids) - 1 - 1 + (P - 1)
    clip = torch.from_numpy(pp.clip_preprocess(img, 224)).bfloat16()[None]
    owl = torch.from_numpy(pp.owl_preprocess(img, 768)).bfloat16()[None]
    direct = vsm.engine.score_batch(clip, owl, ids[None], np.asarray([loc_pos], np.int32))
    assert np.array_equal(boxes.numpy(), direct["pred_boxes"][0])
    assert scores.dtype == torch.bfloat16          # the reference's dtype: ties / thresholds act on bf16-rounded sigmoids
    assert torch.equal(scores, torch.from_numpy(direct["pred_logits"][0]).bfloat16().sigmoid())
    assert np.array_equal(heat.numpy(), vsm.engine.upsample_mask(direct["low_res_masks"][0, 0], 300, 400))
    ref = vsm_oracle.vsm_forward({k: v.float() for k, v in sd.items()}, CFG, clip.float(), owl.float(),
                                 torch.from_numpy(ids.astype(np.int64))[None], LOC)
    assert int(ref["loc_pos"][0]) == loc_pos
    assert np.abs(boxes.numpy() - ref["pred_boxes"][0].numpy()).max() < 1e-2
    # segmentation mode goes through the same fallback
    seg = vsm.inference(img, QUESTION, mode="segmentation")
    assert torch.equal(seg, heat)
    vsm.engine.close()


def test_no_loc_raises_the_references_indexerror(cuda):
    colon, okay, bang = _tid(":"), _tid("Okay"), _tid("!")
    vsm, _ = make_vsm([(colon, okay), (okay, bang), (bang, EOS)])
    img = synthetic_image(320, 320, 3)
    assert vsm.generate_ids(img, QUESTION, 20) == [okay, bang, EOS]
    with pytest.raises(IndexError):
        vsm.inference(img, QUESTION, mode="detection")
    # deferred form: building the batch must not raise, consuming the crop must
    out = vsm.inference_batch([img, img], QUESTION, mode="detection", defer_mismatch=True)
    assert all(isinstance(o, DeferredMismatch) for o in out)
    with pytest.raises(IndexError):
        out[1].resolve()
    vsm.engine.close()


def test_repeated_loc_first_for_boxes_last_for_mask(cuda):
    """Several [LOC] in the output: the reference returns det_result[...][0] (FIRST [LOC]) and pred_mask[-1] (LAST [LOC])
    (visual_search.py:208-225).  Chain ':' -> here -> [LOC] -> ! -> [LOC] -> ! ... (never EOS: 100 generated tokens)."""
    colon, here, bang = _tid(":"), _tid("here"), _tid("!")
    vsm, _ = make_vsm([(colon, here), (here, LOC), (LOC, bang), (bang, LOC)])
    img = synthetic_image(300, 260, 4)
    boxes, scores, heat = vsm.inference(img, QUESTION, mode="detection")
    gen = vsm.fallback_log[0]["generated"]
    assert len(gen) == 100 and gen[:4] == [here, LOC, bang, LOC]
    locs = vsm.fallback_log[0]["loc_positions"]
    assert locs[0] == 1 and locs[-1] == 99
    prompt = _prompt_ids()
    P = CFG.n_img_tokens
    clip = torch.from_numpy(pp.clip_preprocess(img, 224)).bfloat16()[None]
    owl = torch.from_numpy(pp.owl_preprocess(img, 768)).bfloat16()[None]
    ids = np.asarray(prompt + gen[:locs[-1] + 1], np.int32)
    pos = lambda k: len(prompt) + k - 1 + (P - 1)  # noqa: E731
    first = vsm.engine.score_batch(clip, owl, ids[None], np.asarray([pos(locs[0])], np.int32))
    last = vsm.engine.score_batch(clip, owl, ids[None], np.asarray([pos(locs[-1])], np.int32))
    assert np.array_equal(boxes.numpy(), first["pred_boxes"][0])
    assert np.array_equal(heat.numpy(), vsm.engine.upsample_mask(last["low_res_masks"][0, 0], 260, 300))
    # the two [LOC] states differ (predecessor 'here' vs '!'), so this really distinguishes first from last
    assert not np.array_equal(first["low_res_masks"], last["low_res_masks"])
    vsm.engine.close()
