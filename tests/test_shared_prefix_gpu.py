"""VSTAR_F_SHARE_PREFIX: the tokens before <image> (the system prompt: the same text in every crop of a search,
visual_search.py:176-183) go through LLaMA once per call instead of once per crop.

* a crop's record does not depend on the batch it is scored in (alone == inside a batch, bit for bit; B = 1 runs the 128^2 GEMM
  + rope_kernel path, B >= 2 the 256^2 GEMM with the fused RoPE epilogue and row maps);
* against the unshared evaluation the records agree to bf16 rounding noise and pick the same boxes / tokens;
* both agree with the fp32 oracle at the tolerance of the plain path."""
import numpy as np
import pytest
import torch

from oracle import vsm_oracle
from vstar_amd import _lib
from vstar_amd.config import VSMConfig
from vstar_amd.engine import VstarEngine
from vstar_amd.weights import random_state_dict

pytestmark = pytest.mark.gpu


def _inputs(cfg, B, Lp, Lq, seed, same_prefix=True):
    g = torch.Generator().manual_seed(seed)
    I = cfg.clip_image_size
    clip = torch.randn(B, 3, I, I, generator=g).bfloat16()
    owl = torch.randn(B, 3, cfg.owl_image_size, cfg.owl_image_size, generator=g).bfloat16()
    loc_id = cfg.llm_vocab - 1
    L = Lp + 1 + Lq
    ids = torch.randint(3, loc_id - 3, (B, L), generator=g)
    if same_prefix:
        ids[:, :Lp] = ids[0, :Lp]
    ids[:, 0] = 1
    ids[:, Lp] = -200
    ids[:, L - 3] = loc_id
    P = cfg.n_img_tokens
    loc = np.full(B, (L - 3) - 1 + (P - 1), np.int32)
    ver = np.stack([loc - 2, loc - 1, loc, loc + 1], 1).astype(np.int32)
    return clip, owl, ids.numpy().astype(np.int32), loc, ver


@pytest.fixture(scope="module")
def eng(cuda):
    cfg = VSMConfig.tiny(max_batch=6, max_text_len=96, llm_hidden=256, llm_heads=2, llm_mlp=512, llm_layers=3)
    e = VstarEngine(cfg, 0)
    sd = random_state_dict(cfg, seed=11, dtype=torch.bfloat16)
    e.load_state_dict(sd)
    yield e, cfg, sd
    e.close()


@pytest.mark.parametrize("Lp,Lq", [(37, 26), (16, 9), (40, 41)])
def test_shared_prefix_is_batch_invariant_and_matches_unshared(eng, Lp, Lq):
    e, cfg, sd = eng
    B = 6
    clip, owl, ids, loc, ver = _inputs(cfg, B, Lp, Lq, seed=Lp + Lq)
    keys = ("pred_logits", "pred_boxes", "low_res_masks", "tf_argmax")
    shared = e.score_batch(clip, owl, ids, loc, ver, share_prefix=True)
    plain = e.score_batch(clip, owl, ids, loc, ver, share_prefix=False)
    # (1) batch invariance with the flag: crop b alone (128^2 GEMM / rope_kernel path) == crop b in the batch, and a sub-batch
    for sel in ([0], [3], [1, 4], [5, 2, 0]):
        sub = e.score_batch(clip[sel], owl[sel], ids[sel], loc[sel], ver[sel], share_prefix=True)
        for k in keys:
            assert np.array_equal(sub[k], shared[k][sel]), (k, sel)
    # (2) shared vs unshared: a second bf16 evaluation of the same numbers
    assert np.array_equal(shared["tf_argmax"], plain["tf_argmax"])
    for k, tol in (("pred_logits", 2e-2), ("pred_boxes", 1e-2), ("low_res_masks", 6e-2)):
        d = np.abs(shared[k].astype(np.float64) - plain[k].astype(np.float64)).max()
        scale = max(np.abs(plain[k]).max(), 1e-6)
        assert d / scale < tol, (k, d / scale)
    assert np.array_equal(shared["pred_logits"].argmax(1), plain["pred_logits"].argmax(1))
    # (3) the flag is a no-op when the rows do not share their prefix
    clip2, owl2, ids2, loc2, ver2 = _inputs(cfg, B, Lp, Lq, seed=99 + Lp, same_prefix=False)
    a = e.score_batch(clip2, owl2, ids2, loc2, ver2, share_prefix=True)
    b = e.score_batch(clip2, owl2, ids2, loc2, ver2, share_prefix=False)
    for k in keys:
        assert np.array_equal(a[k], b[k]), k


def test_shared_prefix_against_the_oracle(eng):
    e, cfg, sd = eng
    clip, owl, ids, loc, ver = _inputs(cfg, 2, 37, 20, seed=5)
    got = e.score_batch(clip, owl, ids, loc, ver, share_prefix=True)
    plain = e.score_batch(clip, owl, ids, loc, ver, share_prefix=False)
    ref = vsm_oracle.vsm_forward({k: v.float() for k, v in sd.items()}, cfg, clip.float(), owl.float(),
                                 torch.from_numpy(ids.astype(np.int64)), cfg.llm_vocab - 1)
    for k in ("pred_logits", "pred_boxes"):
        r = ref[k].numpy().reshape(got[k].shape)
        e_sh = np.linalg.norm(got[k] - r) / np.linalg.norm(r)
        e_pl = np.linalg.norm(plain[k] - r) / np.linalg.norm(r)
        assert e_sh < max(1.5 * e_pl, 1e-3), (k, e_sh, e_pl)      # no worse than the plain path's distance to fp32


def test_short_prefix_does_not_engage(eng):
    """< 16 tokens before <image> (the golden fixtures' layout): the call is the plain one, bit for bit."""
    e, cfg, _ = eng
    clip, owl, ids, loc, ver = _inputs(cfg, 3, 9, 12, seed=2)
    a = e.score_batch(clip, owl, ids, loc, ver, share_prefix=True)
    b = e.score_batch(clip, owl, ids, loc, ver, share_prefix=False)
    for k in ("pred_logits", "pred_boxes", "low_res_masks", "tf_argmax"):
        assert np.array_equal(a[k], b[k])
