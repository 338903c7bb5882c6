"""CPU stand-in for the HIP engine behind the REAL `vstar_amd.vsm.VSM` class (its prompt building, crop sharding, record
all-gather and post-processing run unchanged): a record is a deterministic function of the crop's preprocessed pixels.
Used by the world-size-2 gloo tests and as `--vsm-factory _fake_vsm:make` for the entry points."""
import warnings

import numpy as np
import torch

from vstar_amd import preprocess as pp
from vstar_amd.engine import VstarEngine


class FakeEngine:
    device = 0

    def __init__(self, max_batch=3):
        from vstar_amd.config import VSMConfig
        self.cfg = VSMConfig.tiny(max_batch=max_batch, max_text_len=96)
        self.calls = []

    def score_batch(self, clip, owl, ids, loc, verify_pos=None, skip_owl=False, sync=True, raw=False):
        from vstar_amd import _lib
        B = clip.shape[0]
        self.calls.append(B)
        rec = np.zeros((B, _lib.RESULT_FLOATS), np.float32)
        for b in range(B):
            # exact integer checksum of the bf16 bit patterns: independent of the summation order / thread count (torchrun runs
            # its workers with OMP_NUM_THREADS=1, and a float sum would differ in the last bits between the two launches)
            bits = lambda t: int(t.contiguous().view(torch.int16).to(torch.int64).sum())  # noqa: E731
            seed = (bits(clip[b]) * 31 + bits(owl[b])) % (2 ** 31)
            g = torch.Generator().manual_seed(seed)
            rec[b, :2304] = (torch.randn(2304, generator=g) * 1.5 - 6).numpy()
            rec[b, 2304:2304 * 5] = torch.rand(2304 * 4, generator=g).numpy()
            low = torch.nn.functional.interpolate(torch.randn(1, 1, 12, 12, generator=g) * 9, (192, 192), mode="bilinear")
            rec[b, 2304 * 5:2304 * 5 + 192 * 192] = low.reshape(-1).numpy()
        return rec if raw else VstarEngine.unpack(rec, 0)

    unpack = staticmethod(VstarEngine.unpack)

    def upsample_mask(self, low, h, w):
        t = torch.from_numpy(np.asarray(low, np.float32)).reshape(1, 1, 192, 192)
        return torch.clamp(torch.nn.functional.interpolate(t, (h, w), mode="bilinear", align_corners=False), min=0)[0, 0].numpy()


def make(args=None, device=0, max_batch=3):
    """--vsm-factory entry: the real VSM wrapper over the fake engine."""
    from vstar_amd.vsm import VSM
    eng = FakeEngine(max_batch)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return VSM(None, engine=eng, tokenizer=pp.SyntheticTokenizer(eng.cfg.llm_vocab), strict_template=False)
