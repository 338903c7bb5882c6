"""Grouped scoring (vstar_vsm_score_grouped): G crops x T prompts that share their first Lp ids — the shared positions run through
LLaMA once per crop, each prompt adds a 32-row suffix block.  Every record must equal the one the plain batch path computes for
that (crop, prompt) pair up to bf16 rounding (the attention tiles the keys differently), and both must sit inside the bf16
noise of the same algorithm (tests/_parity.py) around the fp32 oracle."""
import numpy as np
import pytest
import torch

from _parity import assert_mask_within_bf16_noise, assert_within_bf16_noise, fmt, rel_l2
from oracle import vsm_oracle
from vstar_amd.config import VSMConfig
from vstar_amd.engine import VstarEngine
from vstar_amd.weights import random_state_dict

pytestmark = pytest.mark.gpu


def _case(cfg, G, T, Lp, img_col, lens, seed):
    loc_id = cfg.llm_vocab - 1
    g = torch.Generator().manual_seed(seed)
    I = cfg.clip_image_size
    clip = torch.randn(G, 3, I, I, generator=g).bfloat16()
    owl = torch.randn(G, 3, 768, 768, generator=g).bfloat16()
    prefix = torch.randint(3, loc_id - 3, (Lp,), generator=g).numpy().astype(np.int32)
    prefix[0], prefix[img_col] = 1, -200
    Ls = max(lens)
    suffix = torch.randint(3, loc_id - 3, (G, T, Ls), generator=g).numpy().astype(np.int32)
    loc_in = np.zeros((G, T), np.int32)
    for gi in range(G):
        for t in range(T):
            suffix[gi, t, lens[t] - 2] = loc_id              # "... , [LOC] ."
            loc_in[gi, t] = lens[t] - 3                        # the token in front of [LOC]
    return clip, owl, prefix, suffix, loc_in, Ls, loc_id


@pytest.mark.parametrize("image_size,G,T,Lp,lens", [(224, 2, 3, 9, [7, 12, 5]), (336, 1, 5, 6, [32, 9, 17, 4, 11])])
def test_grouped_equals_per_pair_scoring(cuda, image_size, G, T, Lp, lens):
    cfg = VSMConfig.tiny(clip_image_size=image_size, max_batch=8, max_text_len=48)
    sd = random_state_dict(cfg, seed=13, dtype=torch.bfloat16)
    eng = VstarEngine(cfg, 0)
    eng.load_state_dict(sd)
    clip, owl, prefix, suffix, loc_in, Ls, loc_id = _case(cfg, G, T, Lp, 3, lens, seed=image_size + T)
    P = cfg.n_img_tokens
    Lc = Lp - 1 + P
    ver_in = np.stack([loc_in, np.maximum(loc_in - 1, 0)], axis=-1)
    grp = eng.score_grouped(clip, owl, prefix, suffix, loc_in, ver_in)
    hid_g = eng.debug_read("llm_hidden_loc", G * T * cfg.llm_hidden).reshape(G * T, -1)
    sd32 = {k: v.float() for k, v in sd.items()}
    for gi in range(G):
        for t in range(T):
            n = gi * T + t
            ids = np.concatenate([prefix, suffix[gi, t, :lens[t]]])[None].astype(np.int32)
            loc = np.asarray([Lc + loc_in[gi, t]], np.int32)
            ver = (Lc + ver_in[gi, t])[None].astype(np.int32)
            one = eng.score_batch(clip[gi:gi + 1], owl[gi:gi + 1], ids, loc, verify_pos=ver)
            hid_1 = eng.debug_read("llm_hidden_loc", cfg.llm_hidden)
            tid = torch.from_numpy(ids.astype(np.int64))
            ref = vsm_oracle.vsm_forward(sd32, cfg, clip[gi:gi + 1].float(), owl[gi:gi + 1].float(), tid, loc_id,
                                         verify_pos=torch.from_numpy(ver).long())
            r16 = vsm_oracle.vsm_forward(sd, cfg, clip[gi:gi + 1], owl[gi:gi + 1], tid, loc_id)
            assert int(ref["loc_pos"][0]) == int(loc[0])
            rep = {}
            for name, got, solo in (("llm_hidden_loc", hid_g[n], hid_1), ("pred_logits", grp["pred_logits"][n], one["pred_logits"][0]),
                                    ("pred_boxes", grp["pred_boxes"][n], one["pred_boxes"][0])):
                assert_within_bf16_noise(name, got, ref[name][0].numpy(), r16[name][0].float().numpy(), report=rep)
                noise = rel_l2(r16[name][0].float().numpy(), ref[name][0].numpy())
                assert rel_l2(got, solo) <= 2.0 * noise, (name, rel_l2(got, solo), noise)      # grouped vs plain path: two bf16 runs
            assert_mask_within_bf16_noise(grp["low_res_masks"][n, 0], ref["low_res_masks"][0, 0].numpy(), r16["low_res_masks"][0, 0].float().numpy(),
                                          ref["sam_taps"]["sam_hyper"].numpy(), r16["sam_taps"]["sam_hyper"].float().numpy(),
                                          ref["sam_taps"]["sam_c2"].mean(dim=1).numpy(), r16["sam_taps"]["sam_c2"].float().mean(dim=1).numpy(),
                                          report=rep)
            print(f"\ncrop {gi} prompt {t} (suffix {lens[t]}): {fmt(rep)}")
            # teacher-forced arg-max tokens: equal to the plain path's unless the oracle's own margin is inside the logit noise
            tl = ref["tf_logits"][0].numpy()
            for v in range(2):
                a, b = int(grp["tf_argmax"][n, v]), int(one["tf_argmax"][0, v])
                if a != b:
                    assert abs(tl[v, a] - tl[v, b]) <= 2e-2 * float(tl[v].max() - tl[v].min()), (gi, t, v, a, b)
    # box outputs do not depend on the prompt: identical across the T records of a crop
    for gi in range(G):
        for t in range(1, T):
            assert np.array_equal(grp["pred_boxes"][gi * T], grp["pred_boxes"][gi * T + t])
    eng.close()


def test_grouped_argument_errors(cuda):
    from vstar_amd._lib import VstarError
    cfg = VSMConfig.tiny(max_batch=4, max_text_len=48)
    eng = VstarEngine(cfg, 0)
    eng.load_state_dict(random_state_dict(cfg, seed=1, dtype=torch.bfloat16))
    clip = torch.zeros(1, 3, 224, 224).bfloat16()
    owl = torch.zeros(1, 3, 768, 768).bfloat16()
    pre = np.asarray([1, 5, -200, 7], np.int32)
    with pytest.raises(VstarError, match="G \\* T"):
        eng.score_grouped(clip, owl, pre, np.ones((1, 5, 4), np.int32), np.zeros((1, 5), np.int32))
    with pytest.raises(VstarError, match="-200"):
        eng.score_grouped(clip, owl, np.asarray([1, 5, 6, 7], np.int32), np.ones((1, 2, 4), np.int32), np.zeros((1, 2), np.int32))
    with pytest.raises(VstarError, match="loc_in_suffix"):
        eng.score_grouped(clip, owl, pre, np.ones((1, 2, 4), np.int32), np.full((1, 2), 9, np.int32))
    eng.close()
