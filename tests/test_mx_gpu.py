"""Block-scaled W8A8 activations (vstar_amd/csrc/mx.hpp; BASELINE config 5, round 6) on the MI355X, through the C-ABI's op doors.

The reference has no fp8 path, so the parity target is the oracle's restatement of the scheme (oracle/vsm_oracle.py::mx_fake_quant) —
"parity unpinned" by construction, like the rest of config 5.  What CAN be pinned bit for bit is the internal contract: the producers
that quantise in their epilogues (gate|up GEMM, attention) must write exactly the bytes of `store 16-bit, then vstar_op_quantize_mx`,
and vstar_op_quantize_mx must write exactly the oracle's codes and E8M0 bytes."""
import ctypes
import math

import numpy as np
import pytest
import torch

from oracle import vsm_oracle

pytestmark = pytest.mark.gpu

P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731


def untile_scales(raw: torch.Tensor, rows: int, cols: int) -> torch.Tensor:
    """tile-major scale bytes -> [rows, cols / 32] (mx.hpp::mx_scale_offset)"""
    r = torch.arange(rows).view(-1, 1)
    kb = torch.arange(cols // 32).view(1, -1)
    off = ((kb >> 2) * (rows >> 7) + (r >> 7)) * 512 + ((kb & 3) * 16 + (r & 15)) * 8 + ((r >> 4) & 7)
    return raw.cpu()[off.reshape(-1)].view(rows, cols // 32)


def oracle_codes(x: torch.Tensor):
    dec, e = vsm_oracle.mx_fake_quant(x.cpu())
    scale = torch.ldexp(torch.ones(1), e.to(torch.int32) - 127).repeat_interleave(32, dim=-1)
    return (dec / scale).to(torch.float8_e4m3fn).view(torch.uint8), e, dec


def quantize_mx(lib, x):
    rows, cols = x.shape
    q = torch.zeros(rows, cols, dtype=torch.uint8, device=x.device)
    sc = torch.zeros(lib.vstar_op_mx_scale_bytes(rows, cols), dtype=torch.uint8, device=x.device)
    rc = lib.vstar_op_quantize_mx(None, P(x), P(q), P(sc), rows, cols)
    assert rc == 0, lib.vstar_last_error(None)
    return q, sc


def outlier_rows(g, rows, cols, device):
    x = torch.randn(rows, cols, generator=g, device=device) * (0.05 + 4 * torch.rand(rows, 1, generator=g, device=device))
    x[:, ::97] *= 30.0                                    # outlier channels: what per-token scales suffer from
    x[3, 64:96] = 0.0                                     # an all-zero block
    return x.bfloat16()


@pytest.mark.parametrize("rows,cols", [(128, 128), (384, 4096), (256, 11008)])
def test_quantize_mx_equals_the_oracle(lib, cuda, rows, cols):
    g = torch.Generator(device=cuda).manual_seed(rows + cols)
    x = outlier_rows(g, rows, cols, cuda)
    q, sc = quantize_mx(lib, x)
    codes, e, _ = oracle_codes(x)
    assert torch.equal(untile_scales(sc, rows, cols), e)
    assert torch.equal(q.cpu(), codes)
    # the layout helper the header exports is the one the kernels use
    for (r, kb) in ((0, 0), (17, cols // 64), (rows - 1, cols // 32 - 1)):
        off = lib.vstar_op_mx_scale_offset(r, kb, rows)
        assert int(sc[off]) == int(e[r, kb])


@pytest.mark.parametrize("M,N,K", [(1024, 512, 256), (1280, 4096, 4096), (2560, 4096, 11008)])
def test_gemm_mx_matches_the_fake_quant_oracle(lib, cuda, M, N, K):
    """Block scales applied INSIDE v_mfma_scale_f32_16x16x128_f8f6f4 (per lane: one E8M0 byte per row and 32 k) vs fp32 arithmetic on the
    decoded operands; + residual.  Three repetitions must agree bit for bit (hand-placed waits around a 17th DMA piece per K-tile)."""
    g = torch.Generator(device=cuda).manual_seed(M + N + K)
    A = outlier_rows(g, M, K, cuda)
    W = (torch.randn(N, K, generator=g, device=cuda) / math.sqrt(K) * (0.5 + torch.rand(N, 1, generator=g, device=cuda))).bfloat16()
    res = (torch.randn(M, N, generator=g, device=cuda) * 0.5).bfloat16()
    q, sc = quantize_mx(lib, A)
    outs = []
    for _ in range(3):
        C = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=cuda)
        rc = lib.vstar_op_gemm_mx(None, P(q), P(sc), None, P(W), P(res), P(C), None, None, None, M, N, K, 0, 0, None)
        assert rc == 0, lib.vstar_last_error(None)
        outs.append(C)
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16)) and torch.equal(outs[0].view(torch.int16), outs[2].view(torch.int16))
    dec = vsm_oracle.mx_fake_quant(A.cpu())[0].to(cuda)               # (fp8 casts on the host, the big products on the GPU in fp32)
    wq, sw = (t.to(cuda) for t in vsm_oracle.fp8_fake_quant(W.cpu()))
    ref = ((dec @ wq.T) * sw.T).bfloat16().float() + res.float()
    err = (outs[0].float() - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= 1.2e-2 * scale, (err, scale)             # one bf16 step of the largest outputs, like test_gemm_w8a8_fp8
    # (accuracy: e4m3 is a floating-point format — 2.6 % relative rounding error whatever the scale, over 15 binades — so block
    # scales are neither better nor worse than per-token scales on these rows; what they buy is that a producer can quantise a block
    # the moment it holds its 32 values.  DESIGN.md §9.)


@pytest.mark.parametrize("M,N,K", [(1024, 512, 256), (1280, 22016, 4096)])
def test_gate_up_epilogue_quantises_like_store_then_quantize(lib, cuda, M, N, K):
    g = torch.Generator(device=cuda).manual_seed(M + N + K)
    A = (torch.randn(M, K, generator=g, device=cuda) * (0.2 + 3 * torch.rand(M, 1, generator=g, device=cuda))).bfloat16()
    W = (torch.randn(N, K, generator=g, device=cuda) / math.sqrt(K) * 3).bfloat16()
    C = torch.full((M, N // 2), float("nan"), dtype=torch.bfloat16, device=cuda)
    rc = lib.vstar_op_gemm_fp8(None, P(A), P(W), None, None, P(C), M, N, K, 4, 0, None)
    assert rc == 0, lib.vstar_last_error(None)
    q_ref, sc_ref = quantize_mx(lib, C)
    for _ in range(2):
        q = torch.full((M, N // 2), 0x7F, dtype=torch.uint8, device=cuda)
        sc = torch.zeros_like(sc_ref)
        rc = lib.vstar_op_gemm_fp8_mxout(None, P(A), P(W), P(q), P(sc), M, N, K, 0, None)
        assert rc == 0, lib.vstar_last_error(None)
        assert torch.equal(sc, sc_ref)
        assert torch.equal(q, q_ref)


@pytest.mark.parametrize("B,S,H", [(2, 640, 4), (4, 320, 2), (1, 1152, 3)])
def test_attention_epilogue_quantises_like_store_then_quantize(lib, cuda, B, S, H):
    g = torch.Generator(device=cuda).manual_seed(B * 1000 + S + H)
    qkv = torch.randn(B * S, 3 * H * 128, generator=g, device=cuda).bfloat16()
    out = torch.full((B * S, H * 128), float("nan"), dtype=torch.bfloat16, device=cuda)
    ws = torch.zeros(lib.vstar_op_attention_workspace(B, S, H, 128), dtype=torch.uint8, device=cuda)
    rc = lib.vstar_op_attention(None, P(qkv.clone()), P(out), P(ws), ws.numel(), B, S, H, 128, 1, 0.0)
    assert rc == 0, lib.vstar_last_error(None)
    q_ref, sc_ref = quantize_mx(lib, out)
    q = torch.full((B * S, H * 128), 0x7F, dtype=torch.uint8, device=cuda)
    sc = torch.zeros_like(sc_ref)
    rc = lib.vstar_op_attention_mx(None, P(qkv), P(q), P(sc), B, S, H)
    assert rc == 0, lib.vstar_last_error(None)
    assert torch.equal(sc, sc_ref)
    assert torch.equal(q, q_ref)


@pytest.mark.parametrize("M,N,K", [(1024, 512, 256), (1280, 4096, 4096), (2560, 4096, 11008)])
def test_residual_stream_epilogue_quantises_like_store_then_quantize(lib, cuda, M, N, K):
    """o_proj / down_proj in the fully block-scaled chain: block-scaled A, optional per-row scale, + residual -> the 16-bit rows AND their
    fp8 copy + scales AND sum-of-squares partials.  The 16-bit rows must equal the plain MX consumer's, the fp8 copy must equal
    vstar_op_quantize_mx over them, the partials must give the RMSNorm statistics of the stored rows."""
    g = torch.Generator(device=cuda).manual_seed(M + N + K + 1)
    A = outlier_rows(g, M, K, cuda)
    W = (torch.randn(N, K, generator=g, device=cuda) / math.sqrt(K) * (0.5 + torch.rand(N, 1, generator=g, device=cuda))).bfloat16()
    res = (torch.randn(M, N, generator=g, device=cuda) * 0.5).bfloat16()
    rs = (0.25 + torch.rand(M, generator=g, device=cuda)).float()
    q, sc = quantize_mx(lib, A)
    for row_scale in (None, rs):
        C0 = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=cuda)
        rc = lib.vstar_op_gemm_mx(None, P(q), P(sc), P(row_scale), P(W), P(res), P(C0), None, None, None, M, N, K, 0, 0, None)
        assert rc == 0, lib.vstar_last_error(None)
        C = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=cuda)
        C8 = torch.full((M, N), 0x7F, dtype=torch.uint8, device=cuda)
        cs = torch.zeros(lib.vstar_op_mx_scale_bytes(M, N), dtype=torch.uint8, device=cuda)
        ss = torch.full((M, N // 64), float("nan"), dtype=torch.float32, device=cuda)
        rc = lib.vstar_op_gemm_mx(None, P(q), P(sc), P(row_scale), P(W), P(res), P(C), P(C8), P(cs), P(ss), M, N, K, 0, 0, None)
        assert rc == 0, lib.vstar_last_error(None)
        assert torch.equal(C.view(torch.int16), C0.view(torch.int16))
        q_ref, sc_ref = quantize_mx(lib, C)
        assert torch.equal(cs, sc_ref) and torch.equal(C8, q_ref)
        ref_ss = (C.float() ** 2).view(M, N // 64, 64).sum(-1)
        assert torch.allclose(ss, ref_ss, rtol=1e-5, atol=1e-6)
    if True:      # the per-row scale really multiplies the accumulators (before the residual)
        Ca = torch.zeros(M, N, dtype=torch.bfloat16, device=cuda)
        lib.vstar_op_gemm_mx(None, P(q), P(sc), P(rs), P(W), None, P(Ca), None, None, None, M, N, K, 0, 0, None)
        Cb = torch.zeros(M, N, dtype=torch.bfloat16, device=cuda)
        lib.vstar_op_gemm_mx(None, P(q), P(sc), None, P(W), None, P(Cb), None, None, None, M, N, K, 0, 0, None)
        assert ((Ca.float() - Cb.float() * rs[:, None]).abs().max() <= 1e-2 * Cb.float().abs().max() * rs.max()).item()


@pytest.mark.parametrize("M,N,K", [(1024, 512, 256), (1280, 22016, 4096)])
def test_gate_up_on_block_scaled_input(lib, cuda, M, N, K):
    """gate|up consuming a block-scaled A with the folded norm's row scale: bf16 output vs the fake-quant oracle, and the fp8 output
    equal to store + quantize."""
    g = torch.Generator(device=cuda).manual_seed(M + N + K + 2)
    A = outlier_rows(g, M, K, cuda)
    W = (torch.randn(N, K, generator=g, device=cuda) / math.sqrt(K) * 3).bfloat16()
    rs = (1.0 / A.float().pow(2).mean(-1).sqrt()).float().contiguous()
    q, sc = quantize_mx(lib, A)
    C = torch.full((M, N // 2), float("nan"), dtype=torch.bfloat16, device=cuda)
    rc = lib.vstar_op_gemm_mx(None, P(q), P(sc), P(rs), P(W), None, P(C), None, None, None, M, N, K, 4, 0, None)
    assert rc == 0, lib.vstar_last_error(None)
    dec = vsm_oracle.mx_fake_quant(A.cpu())[0].to(cuda)
    wq, sw = (t.to(cuda) for t in vsm_oracle.fp8_fake_quant(W.cpu()))
    y = ((dec @ wq.T) * sw.T * rs[:, None]).view(M, N // 32, 2, 16)
    ref = (torch.nn.functional.silu(y[:, :, 0].bfloat16().float()).bfloat16().float() * y[:, :, 1].bfloat16().float()).reshape(M, N // 2)
    err = (C.float() - ref).abs().max().item()
    assert err <= 2e-2 * ref.abs().max().item(), (err, ref.abs().max().item())
    q_ref, sc_ref = quantize_mx(lib, C)
    C8 = torch.full((M, N // 2), 0x7F, dtype=torch.uint8, device=cuda)
    cs = torch.zeros_like(sc_ref)
    rc = lib.vstar_op_gemm_mx(None, P(q), P(sc), P(rs), P(W), None, None, P(C8), P(cs), None, M, N, K, 4, 0, None)
    assert rc == 0, lib.vstar_last_error(None)
    assert torch.equal(cs, sc_ref) and torch.equal(C8, q_ref)


def test_mx_doors_refuse_shapes_outside_the_domain(lib, cuda):
    x = torch.zeros(100, 128, dtype=torch.bfloat16, device=cuda)
    assert lib.vstar_op_mx_scale_bytes(100, 128) == 0 and lib.vstar_op_mx_scale_bytes(128, 96) == 0
    q = torch.zeros(100, 128, dtype=torch.uint8, device=cuda)
    assert lib.vstar_op_quantize_mx(None, P(x), P(q), P(q), 100, 128) != 0
    A = torch.zeros(1152, 256, dtype=torch.uint8, device=cuda)         # M % 256 != 0
    sc = torch.zeros(1152 * 8, dtype=torch.uint8, device=cuda)
    W = torch.zeros(256, 256, dtype=torch.bfloat16, device=cuda)
    C = torch.zeros(1152, 256, dtype=torch.bfloat16, device=cuda)
    assert lib.vstar_op_gemm_mx(None, P(A), P(sc), None, P(W), None, P(C), None, None, None, 1152, 256, 256, 0, 0, None) != 0
