"""LLaMA linears at a row count that is NO multiple of 256 (32 crops x 623 positions: a prompt three tokens shorter than the bench's):
the dispatcher's choice (round 6: gemm4w with a ragged last row tile) against the 8-wave gemm256 forced.  TFLOP/s, N(0,1) operands."""
import ctypes, math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vstar_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
M = int(os.environ.get("RAGGED_M", str(32 * 623)))
print(f"M = {M} ({M % 256} rows in the last row tile)")
for name, N, K, epi, res in [("qkv", 12288, 4096, 0, 0), ("o + res", 4096, 4096, 0, 1), ("gate|up silu", 22016, 4096, 4, 0), ("down + res", 4096, 11008, 0, 1)]:
    a = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
    n_out = N // 2 if epi == 4 else N
    c = torch.empty(M, n_out, device=dev, dtype=torch.bfloat16)
    r = torch.randn(M, n_out, device=dev).bfloat16() if res else None
    out = {}
    for tag, flag in (("auto", 0), ("gemm256", _lib.EPI_TILE256)):
        run = lambda: lib.vstar_op_gemm(None, P(a), K, P(w), None, P(r), n_out if res else 0, P(c), n_out, 0, M, N, K, epi | flag | _lib.EPI_NOSYNC)
        best = 1e9
        for _ in range(3):
            for _ in range(2): assert run() == 0, lib.vstar_last_error(None)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): run()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
        out[tag] = (best, lib.vstar_op_gemm_last_tile())
    fl = 2.0 * M * N * K
    print(f"{name:14s} auto: kernel {out['auto'][1]:5d} {fl / out['auto'][0] / 1e9:7.0f} TF/s   gemm256: {fl / out['gemm256'][0] / 1e9:7.0f} TF/s   x{out['gemm256'][0] / out['auto'][0]:.3f}")
