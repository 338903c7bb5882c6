"""--vsm-factory entry for tests/test_two_ranks_one_gpu.py: the REAL engine at the tiny geometry with template-answering weights
(VSM(synthetic_seed=...)), on the device the entry point hands over."""
from vstar_amd.config import VSMConfig
from vstar_amd.vsm import VSM


def make(args=None, device=0):
    return VSM(None, cfg=VSMConfig.tiny(max_batch=4, max_text_len=96), device=device, synthetic_seed=5)
