"""Pins oracle/vsm_oracle.py against the golden vectors produced by the REFERENCE implementation
(oracle/gen_golden.py ran the reference's own model_forward(inference=True) in the build container)."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import vsm_oracle
from oracle.gen_golden import make_inputs
from vstar_amd.config import VSMConfig
from vstar_amd.weights import random_state_dict

GOLDEN = sorted(p for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "tiny*.npz")) if not p.endswith("_bf16.npz"))


def load_case(path):
    z = np.load(path)
    kw = {str(k): int(v) for k, v in zip(z["cfg_keys"], z["cfg_vals"])}
    return z, VSMConfig.tiny(**kw), int(z["weight_seed"]), int(z["loc_id"])


def test_fixtures_present():
    assert len(GOLDEN) >= 3


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_oracle_matches_reference_golden(path):
    z, cfg, wseed, loc_id = load_case(path)
    sd = random_state_dict(cfg, seed=wseed, dtype=torch.float32)
    for i, (seed, L, img_col, loc_col) in enumerate(z["crops"]):
        clip, owl, ids = make_inputs(cfg, int(seed), int(L), int(img_col), int(loc_col), loc_id)
        # the synthetic inputs regenerate bit-identically from their seed
        assert abs(clip.double().sum().item() - z["in_checksum"][i][0]) < 1e-6
        assert np.array_equal(ids[0].numpy().astype(np.int32), z["ids"][i])
        o = vsm_oracle.vsm_forward(sd, cfg, clip, owl, ids, loc_id)

        def close(name, got, tol=2e-5):
            ref = torch.from_numpy(z[name][i])
            err = (got.reshape(ref.shape) - ref).abs().max().item() / max(ref.abs().max().item(), 1e-12)
            assert err < tol, (name, err)

        close("clip_features", o["clip_features"][0])
        close("llm_hidden_loc", o["llm_hidden_loc"][0])
        close("embed_det", o["embed_det"][0])
        close("embed_seg", o["embed_seg"][0])
        close("pred_logits", o["pred_logits"][0, :, 0])
        close("pred_boxes", o["pred_boxes"][0])
        close("low_res_masks", o["low_res_masks"][0, 0])
        assert int(o["loc_pos"][0]) == int(loc_col) - 1 + cfg.n_img_tokens - 1


def test_fp8_fake_quant_helpers():
    """The W8A8 restatement (config 5): codes are exact e4m3 values, scales map the row absmax to 448, and the quantised
    product stays within a few percent of the exact one."""
    g = torch.Generator().manual_seed(0)
    x = torch.randn(7, 512, generator=g) * torch.rand(7, 1, generator=g) * 5
    q, s = vsm_oracle.fp8_fake_quant(x)
    assert torch.allclose(q.abs().amax(dim=1), torch.full((7,), 448.0))
    assert torch.equal(q, q.to(torch.float8_e4m3fn).float())                 # already representable
    assert float(((q * s) - x).abs().max() / x.abs().max()) < 0.07            # 3 mantissa bits
    zero, sz = vsm_oracle.fp8_fake_quant(torch.zeros(2, 16))
    assert torch.equal(zero, torch.zeros(2, 16)) and torch.equal(sz, torch.ones(2, 1))
    w = torch.randn(64, 512, generator=g) / 512 ** 0.5
    y, ref = vsm_oracle.linear_w8a8(x, w), x @ w.T
    assert float((y - ref).norm() / ref.norm()) < 0.05


def test_mx_fake_quant_and_scale_layout(lib):
    """Block-scaled fp8 (round 6, vstar_amd/csrc/mx.hpp): the oracle's restatement picks the SMALLEST power of two that brings a block's
    largest magnitude to <= 448, decodes within e4m3's rounding error, and the tile-major scale layout the library exports is a bijection
    (host arithmetic only: no GPU needed)."""
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(128, 256, generator=g) * torch.rand(128, 1, generator=g) * 5).bfloat16()
    x[3, 64:96] = 0
    dec, e = vsm_oracle.mx_fake_quant(x)
    amax = x.float().view(128, 8, 32).abs().amax(-1)
    ratio = amax / torch.ldexp(torch.ones(1), e.int() - 127)
    assert ratio.max() <= 448 and ratio[amax > 0].min() > 224
    assert e[3, 2] == 0 and dec[3, 64:96].abs().max() == 0
    assert (dec - x.float()).norm() / x.float().norm() < 0.04            # three mantissa bits
    y = vsm_oracle.linear_w8a8_mx(x, torch.randn(64, 256, generator=g).bfloat16())
    assert y.shape == (128, 64) and torch.isfinite(y.float()).all()
    offs = {lib.vstar_op_mx_scale_offset(r, kb, 256) for r in range(256) for kb in range(8)}
    assert offs == set(range(256 * 8)) and lib.vstar_op_mx_scale_bytes(256, 256) == 2048
    assert lib.vstar_op_mx_scale_bytes(100, 256) == 0
