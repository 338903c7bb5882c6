#!/usr/bin/env python
"""gemm256a (hand-scheduled 4-wave AGPR kernel, VSTAR_GEMM256A=1) vs gemm256: bit-identity on several shapes, then timing.
usage: VSTAR_GEMM256A=1 python tools/gemm256a_check.py"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vstar_amd import _lib
assert os.environ.get("VSTAR_GEMM256A") == "1"
lib = _lib.load(); dev = torch.device("cuda:0")
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
def run(a, w, bias, res, c, M, N, K, flags):
    rc = lib.vstar_op_gemm(None, P(a), K, P(w), P(bias), P(res), N, P(c), N, 0, M, N, K, flags)
    assert rc == 0, rc
ok = True
for (M, N, K, hb, hr) in [] if os.environ.get("TIME_ONLY") else [(1024, 256, 128, 0, 0), (1500, 768, 256, 1, 1), (2048, 512, 11008, 0, 0), (20480, 4096, 4096, 0, 1), (4099, 1000, 1024, 1, 0),
                          (8192, 8192, 8192, 0, 0)]:
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g, device=dev).bfloat16()
    npad = (N + 255) // 256 * 256
    w = torch.zeros(npad, K, device=dev, dtype=torch.bfloat16); w[:N] = (torch.randn(N, K, generator=g, device=dev) / K ** 0.5).bfloat16()
    bias = torch.randn(npad, generator=g, device=dev).bfloat16() if hb else None
    res = torch.randn(M, N, generator=g, device=dev).bfloat16() if hr else None
    c0 = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16); c1 = c0.clone()
    run(a, w, bias, res, c0, M, N, K, _lib.EPI_TILE256); t0 = lib.vstar_op_gemm_last_tile()
    for rep in range(3):
        c1.fill_(float("nan"))
        run(a, w, bias, res, c1, M, N, K, 0); t1 = lib.vstar_op_gemm_last_tile()
        torch.cuda.synchronize()
        same = torch.equal(c0.view(torch.int16), c1.view(torch.int16))
        nbad = int((c0.view(torch.int16) != c1.view(torch.int16)).sum())
        ok &= same and t1 == 2560
        if not same or rep == 0: print(f"M{M} N{N} K{K} bias{hb} res{hr}: tiles {t0}/{t1} identical {same} mismatches {nbad} nan {int(torch.isnan(c1.float()).sum())}")
print("ALL_IDENTICAL" if ok else "MISMATCH")
if not ok: sys.exit(1)
if os.environ.get("CHECK_ONLY"): sys.exit(0)
for (name, M, N, K, hr) in [("llama o", 20480, 4096, 4096, 0), ("llama o +res", 20480, 4096, 4096, 1), ("llama down", 20480, 4096, 11008, 0), ("qkv-shape", 20480, 12288, 4096, 0),
                            ("square 8192", 8192, 8192, 8192, 0)]:
    a = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    res = torch.randn(M, N, device=dev).bfloat16() if hr else None
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for label, fl in (("gemm256 ", _lib.EPI_TILE256 | _lib.EPI_NOSYNC), ("gemm256a", _lib.EPI_NOSYNC)):
        for _ in range(3): run(a, w, None, res, c, M, N, K, fl)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run(a, w, None, res, c, M, N, K, fl)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"{name:14s} {label} {ms:7.3f} ms {2.0 * M * N * K / ms / 1e9:8.1f} TFLOP/s")
