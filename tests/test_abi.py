"""The C-ABI library builds, loads on a GPU-less host and exports every symbol include/vstar_hip.h declares."""
import ctypes
import os
import re

import pytest

from vstar_amd import _lib
from vstar_amd.config import CVstarConfig, VSMConfig

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header="vstar_hip.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vstar_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_match_binding_list():
    assert _declared() == sorted(_lib.EXPORTS)


def test_library_exports_every_declared_symbol(lib):
    for name in _declared():
        assert hasattr(lib, name), name


def test_vqa_header_symbols_match_binding_list(lib):
    names = _declared("vstar_vqa.h")
    assert names == sorted(_lib.EXPORTS_VQA)
    for name in names:
        assert hasattr(lib, name), name


def test_vqa_config_layout_and_loud_failure(lib):
    from vstar_amd.config import CVqaConfig, VQAConfig
    assert ctypes.sizeof(CVqaConfig) == 4 * (25 + 8)
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = ctypes.c_void_p()
    c = VQAConfig.tiny().to_c()
    assert lib.vstar_vqa_create(ctypes.byref(c), 0, ctypes.byref(h)) != 0
    c.abi_version = 7
    assert lib.vstar_vqa_create(ctypes.byref(c), 0, ctypes.byref(h)) == -1


def test_result_record_layout():
    # vstar_result is all-gathered byte-for-byte across ranks: its size must match the header
    assert ctypes.sizeof(_lib.VstarResult) == 4 * (2304 + 2304 * 4 + 192 * 192 + 8)
    assert ctypes.sizeof(CVstarConfig) == 4 * (24 + 8)


def test_create_without_gpu_fails_loudly(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = ctypes.c_void_p()
    c = VSMConfig.tiny().to_c()
    rc = lib.vstar_create(ctypes.byref(c), 0, ctypes.byref(h))
    assert rc != 0
    assert b"no CPU fallback" in lib.vstar_last_error(None) or b"device" in lib.vstar_last_error(None)


def test_bad_abi_version_rejected(lib):
    h = ctypes.c_void_p()
    c = VSMConfig.tiny().to_c()
    c.abi_version = 99
    assert lib.vstar_create(ctypes.byref(c), 0, ctypes.byref(h)) == -1
