"""W8A8 (fp8 MFMA) vs bf16 GEMM on the LLaMA-7B linear shapes of the bench (M = 32 x 640 rows)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vstar_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
M = int(os.environ.get("FP8_BENCH_CROPS", "32")) * 640
print(f"{'shape':22s} {'N':>6s} {'K':>6s} {'fp8 ms':>8s} {'fp8 TF/s':>9s} {'bf16 ms':>8s} {'bf16 TF/s':>9s} {'x':>5s}   | W8A8 8-wave gemm256 vs 4-wave gemm4w (round 6), best of 3 interleaved, TFLOP/s")
for name, N, K, epi in [("qkv", 12288, 4096, 0), ("o", 4096, 4096, 0), ("gate_up silu", 22016, 4096, 4), ("down", 4096, 11008, 0)]:
    Kp = (K + 255) // 256 * 256
    a = torch.randn(M, Kp, device=dev).bfloat16(); npad = (N + 255) // 256 * 256
    if Kp != K: a[:, K:] = 0
    w = torch.zeros(npad, Kp, device=dev, dtype=torch.bfloat16); w[:N, :K] = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    n_out = N // 2 if epi == 4 else N
    c = torch.empty(M, n_out, device=dev, dtype=torch.bfloat16)
    ms = ctypes.c_float(0)
    rc = lib.vstar_op_gemm_fp8(None, P(a), P(w), None, None, P(c), M, N, Kp, epi, 10, ctypes.byref(ms))
    assert rc == 0, lib.vstar_last_error(None)
    run = lambda: lib.vstar_op_gemm(None, P(a), Kp, P(w), None, None, 0, P(c), n_out, 0, M, N, Kp, epi | 0x100)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    bms = e0.elapsed_time(e1) / 10
    fl = 2.0 * M * N * K
    best = {0x400: 1e9, 0x800: 1e9}
    for _ in range(3):
        for flag in best:
            t = ctypes.c_float(0)
            assert lib.vstar_op_gemm_fp8(None, P(a), P(w), None, None, P(c), M, N, Kp, epi | flag, 10, ctypes.byref(t)) == 0, lib.vstar_last_error(None)
            best[flag] = min(best[flag], t.value)
    # block-scaled activations (csrc/mx.hpp): o / down CONSUME them (scales inside the MFMA), gate|up PRODUCES them (quantising epilogue)
    mxs = ""
    if Kp == K and M % 256 == 0:
        t = ctypes.c_float(0); bm = 1e9
        if epi == 4:
            q8 = torch.empty(M, n_out, device=dev, dtype=torch.uint8); sc = torch.empty(lib.vstar_op_mx_scale_bytes(M, n_out), device=dev, dtype=torch.uint8)
            for _ in range(3):
                assert lib.vstar_op_gemm_fp8_mxout(None, P(a), P(w), P(q8), P(sc), M, N, K, 10, ctypes.byref(t)) == 0, lib.vstar_last_error(None)
                bm = min(bm, t.value)
            mxs = f"  | mx out {fl / bm / 1e9:7.0f} x{best[0x800] / bm:.3f}"
        elif name != "qkv":
            q8 = torch.empty(M, K, device=dev, dtype=torch.uint8); sc = torch.empty(lib.vstar_op_mx_scale_bytes(M, K), device=dev, dtype=torch.uint8)
            assert lib.vstar_op_quantize_mx(None, P(a), P(q8), P(sc), M, K) == 0
            for _ in range(3):
                assert lib.vstar_op_gemm_mx(None, P(q8), P(sc), None, P(w), None, P(c), None, None, None, M, N, K, 0, 10, ctypes.byref(t)) == 0, lib.vstar_last_error(None)
                bm = min(bm, t.value)
            mxs = f"  | mx in  {fl / bm / 1e9:7.0f} x{best[0x800] / bm:.3f}"
    print(f"{name:22s} {N:6d} {K:6d} {ms.value:8.3f} {fl / ms.value / 1e9:9.1f} {bms:8.3f} {fl / bms / 1e9:9.1f} {bms / ms.value:5.2f}   | "
          f"{fl / best[0x400] / 1e9:7.0f} {fl / best[0x800] / 1e9:7.0f}  x{best[0x400] / best[0x800]:.3f}" + mxs)
