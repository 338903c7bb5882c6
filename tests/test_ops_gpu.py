"""Operator-level parity on the MI355X: each HIP kernel, driven through the C-ABI, against a torch fp32 computation
on the SAME bf16 inputs.  Tolerances: fp32-out GEMM <= 1e-3 relative (fp32 accumulate); bf16 outputs within one bf16
rounding step (2^-8 relative) of the fp32 result plus a small absolute term."""
import ctypes
import math
import os

import pytest
import torch
import torch.nn.functional as F

from vstar_amd import _lib

pytestmark = pytest.mark.gpu

P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731


def _pad_w(w):  # library contract: W allocated to a multiple of 256 rows
    n = (w.shape[0] + 255) // 256 * 256
    out = torch.zeros(n, w.shape[1], dtype=w.dtype, device=w.device)
    out[: w.shape[0]] = w
    return out


def _gemm(lib, a, w, bias=None, res=None, epi=0, f32=False, n=None, tile=0):
    """tile: 0 = dispatcher's choice, 128 / 256 = force that kernel AND assert it is the one that ran."""
    M, K = a.shape
    N = n if n is not None else w.shape[0]
    n_out = N // 2 if epi == _lib.EPI_SILU_MUL else N
    c = torch.full((M, n_out), float("nan"), dtype=torch.float32 if f32 else torch.bfloat16, device=a.device)
    wp = _pad_w(w)
    flag = {0: 0, 128: _lib.EPI_TILE128, 256: _lib.EPI_TILE256, _lib.TILE_4W: _lib.EPI_TILE4W}[tile]
    rc = lib.vstar_op_gemm(None, P(a), K, P(wp), P(bias), P(res), n_out if res is not None else 0, P(c), n_out,
                           1 if f32 else 0, M, N, K, epi | flag)
    assert rc == 0, lib.vstar_last_error(None)
    ran = lib.vstar_op_gemm_last_tile()
    assert ran in (128, 256, 384, _lib.TILE_4W)
    if tile:
        assert ran == tile, f"asked for the {tile}^2 kernel, the {ran}^2 kernel ran"
    torch.cuda.synchronize()
    return c


def _rel(got, ref):
    return ((got.double() - ref.double()).abs().max() / ref.double().abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 128), (200, 130, 192), (1, 514, 768), (777, 1024, 640),
                                   (4096, 4096, 1024), (37, 4, 768)])
def test_gemm_f32_out(lib, cuda, M, N, K):
    g = torch.Generator().manual_seed(M * 7 + N)
    a = torch.randn(M, K, generator=g).bfloat16().to(cuda)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).bfloat16().to(cuda)
    bias = torch.randn(N, generator=g).bfloat16().to(cuda)
    c = _gemm(lib, a, w, bias, f32=True)
    ref = a.float().cpu() @ w.float().cpu().T + bias.float().cpu()
    assert not torch.isnan(c).any()
    assert _rel(c.cpu(), ref) < 1e-3


def test_gemm_detects_transpose(lib, cuda):
    # A = I with an asymmetric W: a swapped C row/col mapping cannot pass
    K = 128
    a = torch.eye(K).bfloat16().to(cuda)
    w = torch.arange(256 * K, dtype=torch.float32).reshape(256, K).remainder(251).bfloat16().to(cuda)
    c = _gemm(lib, a, w, f32=True)
    assert torch.equal(c.cpu(), w.float().cpu().T)


@pytest.mark.parametrize("epi", [_lib.EPI_NONE, _lib.EPI_QUICK_GELU, _lib.EPI_GELU, _lib.EPI_RELU])
def test_gemm_bf16_epilogues(lib, cuda, epi):
    M, N, K = 300, 256, 256
    g = torch.Generator().manual_seed(epi)
    a = torch.randn(M, K, generator=g).bfloat16().to(cuda)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).bfloat16().to(cuda)
    bias = torch.randn(N, generator=g).bfloat16().to(cuda)
    res = torch.randn(M, N, generator=g).bfloat16().to(cuda)
    c = _gemm(lib, a, w, bias, res, epi=epi)
    t = (a.float().cpu() @ w.float().cpu().T + bias.float().cpu()).bfloat16().float()
    if epi == _lib.EPI_QUICK_GELU:
        t = t * torch.sigmoid(1.702 * t)
    elif epi == _lib.EPI_GELU:
        t = F.gelu(t)
    elif epi == _lib.EPI_RELU:
        t = F.relu(t)
    ref = t.bfloat16().float() + res.float().cpu()
    err = (c.float().cpu() - ref).abs()
    assert (err <= ref.abs() * 2 ** -7 + 2e-2).all(), err.max()


def test_gemm_silu_mul(lib, cuda):
    M, F_, K = 130, 64, 128     # F_ = mlp width; packed N = 2*F_ with gate/up interleaved in blocks of 16
    g = torch.Generator().manual_seed(5)
    a = torch.randn(M, K, generator=g).bfloat16().to(cuda)
    gate = (torch.randn(F_, K, generator=g) / math.sqrt(K)).bfloat16()
    up = (torch.randn(F_, K, generator=g) / math.sqrt(K)).bfloat16()
    packed = torch.empty(2 * F_, K, dtype=torch.bfloat16)
    for r in range(2 * F_):
        blk, w = divmod(r, 32)
        packed[r] = gate[blk * 16 + w] if w < 16 else up[blk * 16 + w - 16]
    c = _gemm(lib, a, packed.to(cuda), epi=_lib.EPI_SILU_MUL)
    gf = (a.float().cpu() @ gate.float().T).bfloat16().float()
    uf = (a.float().cpu() @ up.float().T).bfloat16().float()
    ref = F.silu(gf).bfloat16().float() * uf
    err = (c.float().cpu() - ref).abs()
    assert c.shape == (M, F_)
    assert (err <= ref.abs() * 2 ** -7 + 1e-2).all(), err.max()


@pytest.mark.parametrize("rows,cols", [(5, 128), (1000, 1024), (33, 4096), (70, 768), (9, 256), (12, 64)])
def test_layernorm_rmsnorm(lib, cuda, rows, cols):
    g = torch.Generator().manual_seed(rows + cols)
    x = (torch.randn(rows, cols, generator=g) * 2 + 0.3).bfloat16().to(cuda)
    gam = (1 + 0.1 * torch.randn(cols, generator=g)).bfloat16().to(cuda)
    bet = (0.1 * torch.randn(cols, generator=g)).bfloat16().to(cuda)
    y = torch.empty_like(x)
    assert lib.vstar_op_layernorm(None, P(x), P(gam), P(bet), P(y), rows, cols, 1e-5) == 0
    ref = F.layer_norm(x.float().cpu(), (cols,), gam.float().cpu(), bet.float().cpu(), 1e-5)
    assert ((y.float().cpu() - ref).abs() <= ref.abs() * 2 ** -7 + 1e-2).all()
    y2 = torch.empty_like(x)
    assert lib.vstar_op_rmsnorm(None, P(x), P(gam), P(y2), rows, cols, 1e-6) == 0
    xf = x.float().cpu()
    ref2 = gam.float().cpu() * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6)).bfloat16().float()
    assert ((y2.float().cpu() - ref2).abs() <= ref2.abs() * 2 ** -7 + 1e-2).all()


def _attn_ref(qkv, B, S, H, D, causal, theta):
    x = qkv.float().cpu().view(B, S, 3, H, D)
    q, k, v = (x[:, :, i].transpose(1, 2) for i in range(3))
    if theta > 0:
        inv = 1.0 / (theta ** (torch.arange(0, D, 2, dtype=torch.float32) / D))
        f = torch.outer(torch.arange(S, dtype=torch.float32), inv)
        emb = torch.cat([f, f], -1)
        cos, sin = emb.cos().bfloat16().float(), emb.sin().bfloat16().float()
        rot = lambda t: torch.cat([-t[..., D // 2:], t[..., : D // 2]], -1)  # noqa: E731
        q = ((q * cos).bfloat16().float() + (rot(q) * sin).bfloat16().float()).bfloat16().float()
        k = ((k * cos).bfloat16().float() + (rot(k) * sin).bfloat16().float()).bfloat16().float()
    s = (q @ k.transpose(-1, -2)) / math.sqrt(D)
    if causal:
        s = s + torch.full((S, S), float("-inf")).triu(1)
    o = torch.softmax(s, -1) @ v
    return o.transpose(1, 2).reshape(B * S, H * D)


@pytest.mark.parametrize("B,S,H,D,causal,theta", [
    (2, 257, 2, 64, 0, 0.0), (1, 577, 3, 64, 0, 0.0), (1, 2305, 2, 64, 0, 0.0), (2, 320, 2, 128, 1, 10000.0),
    (1, 640, 2, 128, 1, 10000.0), (3, 33, 1, 128, 1, 10000.0), (1, 100, 2, 64, 1, 0.0), (1, 64, 1, 128, 0, 0.0)])
def test_attention(lib, cuda, B, S, H, D, causal, theta):
    g = torch.Generator().manual_seed(S + D)
    qkv = torch.randn(B * S, 3 * H * D, generator=g).bfloat16()
    ref = _attn_ref(qkv, B, S, H, D, causal, theta)
    dq = qkv.to(cuda)
    out = torch.full((B * S, H * D), float("nan"), dtype=torch.bfloat16, device=cuda)
    ws_bytes = lib.vstar_op_attention_workspace(B, S, H, D)
    ws = torch.zeros(ws_bytes, dtype=torch.uint8, device=cuda)
    rc = lib.vstar_op_attention(None, P(dq), P(out), P(ws), ws_bytes, B, S, H, D, causal, theta)
    assert rc == 0, lib.vstar_last_error(None)
    o = out.float().cpu()
    assert not torch.isnan(o).any()
    # P is rounded to bf16 before the PV product and the output is bf16: ~1e-2 absolute on unit-variance V
    print(f"ATTN_ERR plain B{B} S{S} H{H} D{D} c{causal} rope{int(theta > 0)}: abs {(o - ref).abs().max().item():.3e} rel {_rel(o, ref):.3e}")
    # gates = 2 x the measured error of this kernel (profiles/r03_attention_op_error.txt: abs <= 8.2e-3, rel <= 5.3e-3 over these
    # shapes; spiked key <= 1.44e-2 abs): a 2 x regression fails
    assert (o - ref).abs().max().item() < 1.7e-2
    assert _rel(o, ref) < 1.1e-2
    # a spiked key (rule: force the online-softmax rescale branch): one huge q.k at a late tile
    if S >= 64:
        x = qkv.clone().view(B, S, 3, H, D)
        x[0, S - 1, 0, 0] = 4.0
        x[0, S - 2, 1, 0] = 4.0
        ref2 = _attn_ref(x.view(B * S, -1), B, S, H, D, causal, 0.0)
        dq2 = x.view(B * S, -1).to(cuda)
        rc = lib.vstar_op_attention(None, P(dq2), P(out), P(ws), ws_bytes, B, S, H, D, causal, 0.0)
        assert rc == 0
        print(f"ATTN_ERR spike B{B} S{S} H{H} D{D} c{causal}: abs {(out.float().cpu() - ref2).abs().max().item():.3e}")
        assert (out.float().cpu() - ref2).abs().max().item() < 2.9e-2


@pytest.mark.parametrize("S,H,D,causal", [(577, 2, 64, 0), (640, 2, 128, 1), (200, 1, 128, 1)])
@pytest.mark.parametrize("profile", ["ascending", "descending", "all_very_negative", "sawtooth"])
def test_attention_softmax_dynamic_range(lib, cuda, S, H, D, causal, profile):
    """The kernel subtracts a STALE running maximum (handed to the QK^T MFMA as its C operand) and only re-centres on a slow
    path: a wave's first sub-tile, and whenever a probability outgrows 2^16.  These score profiles drive that path hard:
    scores that climb by e^90 along the keys (re-centring in every tile), that fall by as much (later tiles underflow),
    that sit at -300 everywhere (the first sub-tile must anchor m to a true maximum or the row sum underflows), and a sawtooth."""
    g = torch.Generator().manual_seed(S + D + len(profile))
    x = (0.1 * torch.randn(1, S, 3, H, D, generator=g))
    j = torch.arange(S, dtype=torch.float32)
    ramp = {"ascending": 90.0 * j / S, "descending": 90.0 * (1 - j / S), "all_very_negative": torch.full((S,), -300.0),
            "sawtooth": 40.0 * ((j % 97) / 97.0) + 30.0 * (j // 97 % 2)}[profile]
    x[0, :, 0, :, 0] = math.sqrt(D)                        # q[:, 0] = sqrt(D): score(i, j) = k[j, 0] + small noise
    x[0, :, 1, :, 0] = ramp[:, None]
    x[0, :, 2] = torch.randn(S, H, D, generator=g)         # V unit variance
    qkv = x.view(S, -1).bfloat16()
    ref = _attn_ref(qkv, 1, S, H, D, causal, 0.0)
    out = torch.full((S, H * D), float("nan"), dtype=torch.bfloat16, device=cuda)
    ws_bytes = lib.vstar_op_attention_workspace(1, S, H, D)
    ws = torch.zeros(ws_bytes, dtype=torch.uint8, device=cuda)
    rc = lib.vstar_op_attention(None, P(qkv.to(cuda)), P(out), P(ws), ws_bytes, 1, S, H, D, causal, 0.0)
    assert rc == 0, lib.vstar_last_error(None)
    o = out.float().cpu()
    assert torch.isfinite(o).all()
    # the pre-scaled q (bf16) moves a score of magnitude 90 by ~0.15; a CPU emulation of the kernel's rounding points lands at
    # 5e-3 relative on these profiles
    print(f"ATTN_ERR range S{S} D{D} c{causal} {profile}: abs {(o - ref).abs().max().item():.3e} rel {_rel(o, ref):.3e}")
    # measured: abs <= 8.2e-3, rel <= 6.1e-3 (profiles/r03_attention_op_error.txt); gate = 2 x
    assert (o - ref).abs().max().item() < 1.7e-2
    assert _rel(o, ref) < 1.25e-2


# ---- the 256x256 8-phase kernel (M >= 1024, N >= 256, K % 128 == 0): tails, long K, epilogues, race screen ----
@pytest.mark.parametrize("M,N,K", [(1024, 256, 128), (1500, 768, 256), (1030, 300, 384), (2048, 512, 11008),
                                   (20480, 1024, 4096), (1300, 4096, 1024)])
def test_gemm256_f32_out(lib, cuda, M, N, K):
    g = torch.Generator(device=cuda).manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g, device=cuda).bfloat16()
    w = (torch.randn(N, K, generator=g, device=cuda) / math.sqrt(K)).bfloat16()
    bias = torch.randn(N, generator=g, device=cuda).bfloat16()
    c = _gemm(lib, a, w, bias, f32=True, tile=256)
    ref = a.float() @ w.float().T + bias.float()          # device fp32 GEMM as the large-shape checker
    assert not torch.isnan(c).any()
    assert _rel(c, ref) < 1e-3
    # every element, not just the max: catches a single stale LDS tile
    assert ((c - ref).abs() <= 1e-3 * ref.abs().max()).all()


def test_gemm256_transpose_and_rowmajor(lib, cuda):
    K, N = 256, 512
    a = torch.zeros(1024, K)
    a[:K] = torch.eye(K)
    w = torch.arange(N * K, dtype=torch.float32).reshape(N, K).remainder(251)
    c = _gemm(lib, a.bfloat16().to(cuda), w.bfloat16().to(cuda), f32=True, tile=256)
    assert torch.equal(c[:K].cpu(), w.bfloat16().float().T)
    assert (c[K:] == 0).all()


@pytest.mark.parametrize("epi", [_lib.EPI_NONE, _lib.EPI_QUICK_GELU, _lib.EPI_RELU])
def test_gemm256_bf16_epilogues(lib, cuda, epi):
    M, N, K = 1200, 384, 128
    g = torch.Generator().manual_seed(40 + epi)
    a = torch.randn(M, K, generator=g).bfloat16().to(cuda)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).bfloat16().to(cuda)
    bias = torch.randn(N, generator=g).bfloat16().to(cuda)
    res = torch.randn(M, N, generator=g).bfloat16().to(cuda)
    c = _gemm(lib, a, w, bias, res, epi=epi, tile=256)
    t = (a.float().cpu() @ w.float().cpu().T + bias.float().cpu()).bfloat16().float()
    if epi == _lib.EPI_QUICK_GELU:
        t = t * torch.sigmoid(1.702 * t)
    elif epi == _lib.EPI_RELU:
        t = F.relu(t)
    ref = t.bfloat16().float() + res.float().cpu()
    err = (c.float().cpu() - ref).abs()
    assert (err <= ref.abs() * 2 ** -7 + 2e-2).all(), err.max()


def test_gemm256_silu_mul(lib, cuda):
    M, F_, K = 1100, 256, 256
    g = torch.Generator().manual_seed(6)
    a = torch.randn(M, K, generator=g).bfloat16().to(cuda)
    gate = (torch.randn(F_, K, generator=g) / math.sqrt(K)).bfloat16()
    up = (torch.randn(F_, K, generator=g) / math.sqrt(K)).bfloat16()
    idx = torch.arange(2 * F_)
    blk, wi = idx // 32, idx % 32
    packed = torch.where((wi < 16)[:, None], gate[(blk * 16 + wi.clamp(max=15))], up[(blk * 16 + (wi - 16).clamp(min=0))])
    c = _gemm(lib, a, packed.to(cuda), epi=_lib.EPI_SILU_MUL, tile=256)
    gf = (a.float().cpu() @ gate.float().T).bfloat16().float()
    uf = (a.float().cpu() @ up.float().T).bfloat16().float()
    ref = F.silu(gf).bfloat16().float() * uf
    assert c.shape == (M, F_)
    assert ((c.float().cpu() - ref).abs() <= ref.abs() * 2 ** -7 + 1e-2).all()


def test_gemm256_race_screen(lib, cuda):
    """The DMA ring is guarded only by counted vmcnt waits and barriers: repeat a many-tile GEMM and demand
    bit-identical results every time while the chip is busy (2 waves of tiles per CU)."""
    M, N, K = 8192, 4096, 2048
    g = torch.Generator(device=cuda).manual_seed(77)
    a = torch.randn(M, K, generator=g, device=cuda).bfloat16()
    w = (torch.randn(N, K, generator=g, device=cuda) / math.sqrt(K)).bfloat16()
    first = _gemm(lib, a, w, f32=True, tile=256)
    ref = a.float() @ w.float().T
    assert ((first - ref).abs() <= 1e-3 * ref.abs().max()).all()
    for _ in range(10):
        again = _gemm(lib, a, w, f32=True, tile=256)
        assert torch.equal(again, first)


def _pack_gate_up(gate, up):
    F_ = gate.shape[0]
    idx = torch.arange(2 * F_)
    blk, wi = idx // 32, idx % 32
    return torch.where((wi < 16)[:, None], gate[(blk * 16 + wi.clamp(max=15))], up[(blk * 16 + (wi - 16).clamp(min=0))])


def test_gemm_tile_override_contract(lib, cuda):
    """The per-call tile override is honoured or refused, never silently re-routed; the dispatcher's own choice is observable."""
    a = torch.randn(300, 128).bfloat16().to(cuda)
    w = torch.randn(256, 128).bfloat16().to(cuda)
    c = torch.empty(300, 256, dtype=torch.bfloat16, device=cuda)
    rc = lib.vstar_op_gemm(None, P(a), 128, P(w), None, None, 0, P(c), 256, 0, 300, 256, 128, _lib.EPI_TILE256)
    assert rc != 0 and b"256x256" in lib.vstar_last_error(None)          # M < 1024: outside the 256^2 kernel's domain
    rc = lib.vstar_op_gemm(None, P(a), 128, P(w), None, None, 0, P(c), 256, 0, 300, 256, 128, _lib.EPI_TILE256 | _lib.EPI_TILE128)
    assert rc != 0
    _gemm(lib, a, w, tile=128)
    # dispatcher defaults: a full grid of interior 256^2 tiles goes to the 4-wave kernel (round 6), one with an M tail stays on the
    # 8-wave 256^2 kernel, an under-filled one moves to 128^2
    big_a = torch.randn(20480, 256).bfloat16().to(cuda)
    big_w = torch.randn(4096, 256).bfloat16().to(cuda)
    _gemm(lib, big_a, big_w)
    assert lib.vstar_op_gemm_last_tile() == (_lib.TILE_4W if os.environ.get("VSTAR_GEMM4W", "1") != "0" else 256)
    _gemm(lib, big_a[:20400], big_w)
    assert lib.vstar_op_gemm_last_tile() == 256
    _gemm(lib, big_a[:1200], big_w[:384])
    assert lib.vstar_op_gemm_last_tile() == 128


@pytest.mark.parametrize("M,N,K,epi,use_bias,use_res,f32,split", [
    (18464, 1024, 256, 0, True, True, False, True),      # CLIP out-proj / fc2 form: 292 tiles = 1 round + 36 on 256 CUs -> split
    (73760, 2304, 128, 1, True, False, False, True),     # OWL qkv width with QUICK_GELU: 10 rounds + 41 tiles -> split
    (18464, 3072, 128, 0, False, True, False, False),    # CLIP qkv width: the tail would need two rounds of 128^2 tiles -> no split
    (18464, 1024, 256, 0, True, False, True, True),      # fp32 output through the split
])
def test_gemm_ragged_round_split_is_bit_identical(lib, cuda, M, N, K, epi, use_bias, use_res, f32, split):
    """Dispatcher default on a 256-CU device: a thin last round of 256^2 tiles is replaced by 128^2 tiles over the trailing
    rows (vstar_op_gemm_last_tile() == 384).  Same K order in both kernels: the result equals the forced single-kernel one."""
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).bfloat16().to(cuda)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).bfloat16().to(cuda)
    bias = torch.randn(N, generator=g).bfloat16().to(cuda) if use_bias else None
    res = torch.randn(M, N, generator=g).bfloat16().to(cuda) if use_res else None
    c = _gemm(lib, a, w, bias, res, epi=epi, f32=f32)
    ran = lib.vstar_op_gemm_last_tile()
    c256 = _gemm(lib, a, w, bias, res, epi=epi, f32=f32, tile=256)
    assert torch.equal(c, c256)
    if torch.cuda.get_device_properties(0).multi_processor_count == 256:
        assert ran == (384 if split else 256), ran


@pytest.mark.parametrize("M,N,K,epi,use_bias,use_res,f32", [
    (1030, 300, 384, 0, True, False, True),        # M and N tails, fp32 out
    (1500, 768, 256, 1, True, False, False),       # QUICK_GELU + bias (ViT fc1 form)
    (1300, 4096, 1024, 0, False, True, False),     # residual (o_proj / fc2 form)
    (2048, 512, 11008, 0, False, True, False),     # long K (down_proj)
    (1100, 512, 256, 4, False, False, False),      # SiLU(gate)*up
    (1200, 384, 128, 3, True, True, False),        # RELU + bias + residual, N tail
    (1056, 514, 768, 0, True, False, True),        # fused class-head width (N = 514)
    (1280, 4096, 4096, 0, False, True, False),     # 320 tiles of 128^2, long K: the 128 x 256 loader-wave tile (o_proj, 2 crops)
    (1280, 8192, 2048, 4, False, False, False),    # the same tile with SiLU(gate)*up
    (1150, 6912, 2048, 0, True, False, True),      # ... fp32 output, bias, M tail
])
def test_gemm128_equals_gemm256(lib, cuda, M, N, K, epi, use_bias, use_res, f32):
    """Both kernels accumulate K in the same order with the same epilogue arithmetic: forced onto the same operands they
    must agree BIT FOR BIT (this is what lets the dispatcher pick either by grid fill without changing results)."""
    g = torch.Generator().manual_seed(M * 3 + N + K + epi)
    a = torch.randn(M, K, generator=g).bfloat16().to(cuda)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).bfloat16()
    if epi == _lib.EPI_SILU_MUL:
        w = _pack_gate_up(w[: N // 2], w[N // 2:])
    w = w.to(cuda)
    n_out = N // 2 if epi == _lib.EPI_SILU_MUL else N
    bias = torch.randn(N, generator=g).bfloat16().to(cuda) if use_bias else None
    res = torch.randn(M, n_out, generator=g).bfloat16().to(cuda) if use_res else None
    c256 = _gemm(lib, a, w, bias, res, epi=epi, f32=f32, tile=256)
    c128 = _gemm(lib, a, w, bias, res, epi=epi, f32=f32, tile=128)
    assert not torch.isnan(c256.float()).any()
    assert torch.equal(c256, c128)


@pytest.mark.parametrize("M,N,K,epi,use_bias,use_res", [
    (1024, 256, 128, 0, False, False),          # the smallest shape of the domain: two K-tiles, no loop iteration
    (1024, 512, 256, 3, True, True),            # RELU + bias + residual, one loop iteration
    (1536, 768, 384, 1, True, False),           # QUICK_GELU + bias, K/64 = 6
    (1280, 4096, 4096, 0, False, True),         # o_proj form: residual
    (2048, 512, 11008, 0, False, True),         # long K (down_proj)
    (1024, 8192, 2048, 4, False, False),        # SiLU(gate)*up
    (20480, 4096, 4096, 0, True, True),         # the bench batch's o_proj rows: 1280 tiles, five per workgroup (persistent loop)
    (1300, 512, 4096, 0, True, True),           # ragged last row tile (round 6: prompts whose rows are no multiple of 256): 20 of 256 rows
    (2100, 768, 4096, 1, True, False),          # ragged + QUICK_GELU
    (1111, 1024, 4096, 4, False, False),        # ragged (not even a multiple of 16) + SiLU(gate)*up
])
def test_gemm4w_equals_gemm256(lib, cuda, M, N, K, epi, use_bias, use_res):
    """The 4-wave / AGPR 256^2 kernel (gemm4w.hip, hand-scheduled K loop) against the 8-wave gemm256 on the same operands: same k
    order, same epilogue arithmetic -> BIT-identical.  Five repetitions each: the loop's LDS hand-offs are ordered by counted
    vmcnt + barriers only, a misplaced wait shows up as a rare wrong tile."""
    g = torch.Generator(device=cuda).manual_seed(M * 3 + N + K + epi)
    a = torch.randn(M, K, generator=g, device=cuda).bfloat16()
    w = (torch.randn(N, K, generator=g, device=cuda) / math.sqrt(K)).bfloat16()
    if epi == _lib.EPI_SILU_MUL:
        w = _pack_gate_up(w[: N // 2].cpu(), w[N // 2:].cpu()).to(cuda)
    n_out = N // 2 if epi == _lib.EPI_SILU_MUL else N
    bias = torch.randn(N, generator=g, device=cuda).bfloat16() if use_bias else None
    res = torch.randn(M, n_out, generator=g, device=cuda).bfloat16() if use_res else None
    c256 = _gemm(lib, a, w, bias, res, epi=epi, tile=256)
    assert not torch.isnan(c256.float()).any()
    for rep in range(5):
        c4w = _gemm(lib, a, w, bias, res, epi=epi, tile=_lib.TILE_4W)
        bad = (c4w.view(torch.int16) != c256.view(torch.int16)).sum().item()
        assert bad == 0, (rep, bad)


@pytest.mark.parametrize("M,N,K,epi", [(1024, 512, 256, 0), (2048, 4096, 4096, 0), (1280, 8192, 1024, 4), (20480, 4096, 11008, 0),
                                       (1300, 512, 4096, 0), (2100, 1024, 4096, 4)])      # the last two: ragged last row tile
def test_gemm4w_w8a8_equals_gemm256_w8a8(lib, cuda, M, N, K, epi):
    """Round 6: the W8A8 instantiation of the 4-wave kernel (v_mfma_scale_f32_16x16x128_f8f6f4 in its own generated K loop: two A
    fragment sets, W fragments refilled on a rolling basis) against gemm256's on the same quantised operands: same k order per
    MFMA, same dequantisation and epilogue -> BIT-identical; three repetitions (hand-placed waits)."""
    g = torch.Generator(device=cuda).manual_seed(M + N + K + epi)
    n_out = N // 2 if epi == 4 else N
    A = (torch.randn(M, K, generator=g, device=cuda) * (0.2 + 3 * torch.rand(M, 1, generator=g, device=cuda))).bfloat16()
    W = (torch.randn(N, K, generator=g, device=cuda) / math.sqrt(K)).bfloat16()
    bias = (torch.randn(N, generator=g, device=cuda) * 0.1).bfloat16() if epi == 0 else None
    res = (torch.randn(M, n_out, generator=g, device=cuda) * 0.5).bfloat16() if epi == 0 else None
    outs = []
    for flag in (_lib.EPI_TILE256, _lib.EPI_TILE4W, _lib.EPI_TILE4W, _lib.EPI_TILE4W):
        C = torch.full((M, n_out), float("nan"), dtype=torch.bfloat16, device=cuda)
        rc = lib.vstar_op_gemm_fp8(None, P(A), P(W), P(bias), P(res), P(C), M, N, K, epi | flag, 0, None)
        assert rc == 0, lib.vstar_last_error(None)
        torch.cuda.synchronize()
        assert lib.vstar_op_gemm_last_tile() == (256 if flag == _lib.EPI_TILE256 else _lib.TILE_4W)
        outs.append(C)
    assert not torch.isnan(outs[0].float()).any()
    for c in outs[1:]:
        assert torch.equal(c.view(torch.int16), outs[0].view(torch.int16))


@pytest.mark.parametrize("M,N,K", [(1024, 4096, 256), (2048, 512, 1024), (1300, 512, 4096)])
def test_gemm4w_row_scale_and_statistics(lib, cuda, M, N, K):
    """Folded-norm row scale in, sum-of-squares partials out (the LLaMA o_proj / down / q|k|v forms): bit-identical to gemm256."""
    g = torch.Generator(device=cuda).manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g, device=cuda).bfloat16()
    w = (torch.randn(N, K, generator=g, device=cuda) / math.sqrt(K)).bfloat16()
    res = (3 * torch.randn(M, N, generator=g, device=cuda)).bfloat16()
    rs = (0.5 + torch.rand(M, generator=g, device=cuda)).float()
    c256, s256 = _gemm_norm(lib, a, w, res, tile=256, row_scale=rs, want_sumsq=True)
    c4w, s4w = _gemm_norm(lib, a, w, res, tile=_lib.TILE_4W, row_scale=rs, want_sumsq=True)
    torch.cuda.synchronize()
    assert torch.equal(c256, c4w) and torch.equal(s256, s4w)
    c256, _ = _gemm_norm(lib, a, w, None, epi=_lib.EPI_NONE, tile=256, row_scale=rs)
    c4w, _ = _gemm_norm(lib, a, w, None, epi=_lib.EPI_NONE, tile=_lib.TILE_4W, row_scale=rs)
    torch.cuda.synchronize()
    assert torch.equal(c256, c4w)


def test_gemm4w_refuses_shapes_outside_its_domain(lib, cuda):
    a = torch.randn(1100, 256, device=cuda).bfloat16()           # M % 256 != 0 with a short K: gemm256's shape (a ragged last row tile is
    w = torch.randn(256, 256, device=cuda).bfloat16()            # taken from K = 4096 up only)
    c = torch.empty(1100, 256, device=cuda, dtype=torch.bfloat16)
    rc = lib.vstar_op_gemm(None, P(a), 256, P(_pad_w(w)), None, None, 0, P(c), 256, 0, 1100, 256, 256, _lib.EPI_TILE4W)
    assert rc != 0


@pytest.mark.parametrize("M", [1030, 1279])
def test_gemm4w_ragged_tile_touches_nothing_past_the_matrix(lib, cuda, M):
    """The last row tile of a ragged launch: rows >= M of the output, the residual and the statistics buffer stay untouched (the
    epilogue skips them lane by lane), and A is read to the end of row M - 1 only — here it ends at the end of its allocation's
    used part, with NaNs behind it that would poison the tile's valid rows if the DMA of the missing rows were not zero-filled."""
    N, K = 512, 4096
    g = torch.Generator(device=cuda).manual_seed(M)
    abuf = torch.full((M + 256, K), float("nan"), dtype=torch.bfloat16, device=cuda)
    abuf[:M] = torch.randn(M, K, generator=g, device=cuda).bfloat16()
    w = (torch.randn(N, K, generator=g, device=cuda) / math.sqrt(K)).bfloat16()
    cbuf = torch.full((M + 256, N), 123.0, dtype=torch.bfloat16, device=cuda)
    rbuf = torch.randn(M + 256, N, generator=g, device=cuda).bfloat16()
    rc = lib.vstar_op_gemm(None, P(abuf), K, P(_pad_w(w)), None, P(rbuf), N, P(cbuf), N, 0, M, N, K, _lib.EPI_TILE4W)
    assert rc == 0, lib.vstar_last_error(None)
    assert lib.vstar_op_gemm_last_tile() == _lib.TILE_4W
    torch.cuda.synchronize()
    assert (cbuf[M:] == 123.0).all()
    ref = _gemm(lib, abuf[:M].contiguous(), w, None, rbuf[:M].contiguous(), tile=256)
    assert not torch.isnan(cbuf[:M].float()).any()
    assert torch.equal(cbuf[:M].view(torch.int16), ref.view(torch.int16))


@pytest.mark.parametrize("name,M,N,K,epi,use_bias,use_res", [
    ("clip_qkv", 18464, 3072, 1024, 0, True, False),          # 32 crops x 577 tokens: M tail of 32 rows
    ("clip_fc1", 18464, 4096, 1024, 1, True, False),          # QUICK_GELU + bias, short K
    ("clip_fc2", 18464, 1024, 4096, 0, True, True),           # bias + residual
    ("owl_qkv", 73760, 2304, 768, 0, True, False),            # 32 x 2305 tokens: M tail of 32 rows, N = 9 tiles
    ("owl_fc1", 73760, 3072, 768, 1, True, False),
    ("llm_gate_up", 20480, 22016, 4096, 4, False, False),     # SiLU(gate)*up, N = 86 tiles
    ("llm_down", 20480, 4096, 11008, 0, False, True),         # K = 11008 + residual
    ("llm_o", 20480, 4096, 4096, 0, False, True),
])
def test_gemm256_bench_shapes(lib, cuda, name, M, N, K, epi, use_bias, use_res):
    """The shapes the headline bench actually launches (B = 32, I = 336), each forced onto the 256^2 kernel and checked on
    every element against a device fp32 GEMM + the reference's bf16 rounding points."""
    g = torch.Generator(device=cuda).manual_seed(len(name) + M + N)
    a = torch.randn(M, K, generator=g, device=cuda).bfloat16()
    w = (torch.randn(N, K, generator=g, device=cuda) / math.sqrt(K)).bfloat16()
    n_out = N // 2 if epi == _lib.EPI_SILU_MUL else N
    bias = torch.randn(N, generator=g, device=cuda).bfloat16() if use_bias else None
    res = torch.randn(M, n_out, generator=g, device=cuda).bfloat16() if use_res else None
    wk = _pack_gate_up(w[: N // 2].cpu(), w[N // 2:].cpu()).to(cuda) if epi == _lib.EPI_SILU_MUL else w
    c = _gemm(lib, a, wk, bias, res, epi=epi, tile=256).float()
    del wk
    if epi == _lib.EPI_SILU_MUL:
        gf = (a.float() @ w[: N // 2].float().T).bfloat16().float()
        uf = (a.float() @ w[N // 2:].float().T).bfloat16().float()
        ref = F.silu(gf).bfloat16().float() * uf
        del gf, uf
    else:
        t = a.float() @ w.float().T
        if bias is not None:
            t += bias.float()
        t = t.bfloat16().float()
        if epi == _lib.EPI_QUICK_GELU:
            t = t * torch.sigmoid(1.702 * t)
        ref = t.bfloat16().float() + res.float() if res is not None else t
    assert not torch.isnan(c).any()
    err = (c - ref).abs()
    # bf16 output = one rounding (2^-8 relative); SiLU(g)*u rounds g, u and SiLU(g) first, and a 1e-6 difference in the fp32
    # accumulation order can flip any of them by one ulp on a rounding boundary (a handful of the 2e8 outputs): 4 x 2^-8
    tol = ref.abs() * (2 ** -6 if epi in (_lib.EPI_SILU_MUL, _lib.EPI_QUICK_GELU) else 2 ** -7) + 2e-2
    bad = (err > tol).sum().item()
    assert bad == 0, (name, bad, err.max().item())


@pytest.mark.parametrize("M,N,K,epi", [(1024, 512, 256, 0), (1300, 768, 1024, 0), (2048, 1024, 4096, 4), (1056, 300, 512, 0)])
def test_gemm_w8a8_fp8(cuda, lib, M, N, K, epi):
    """W8A8 GEMM on the fp8 MFMA (BASELINE config 5) vs torch on the SAME quantised operands: per-token / per-output-channel
    absmax/448 scales, OCP e4m3 round-to-nearest-even, fp32 accumulation, then the usual epilogue."""
    g = torch.Generator().manual_seed(M + N + K)
    Npad = (N + 255) // 256 * 256
    n_out = N // 2 if epi == 4 else N
    A = (torch.randn(M, K, generator=g) * torch.rand(M, 1, generator=g) * 3).bfloat16()
    W = torch.zeros(Npad, K, dtype=torch.bfloat16)
    W[:N] = (torch.randn(N, K, generator=g) / K ** 0.5 * (0.5 + torch.rand(N, 1, generator=g))).bfloat16()
    bias = (torch.randn(Npad, generator=g) * 0.1).bfloat16() if epi == 0 else None
    res = (torch.randn(M, n_out, generator=g) * 0.5).bfloat16() if epi == 0 else None
    dA, dW = A.cuda(), W.cuda()
    dB = bias.cuda() if bias is not None else None
    dR = res.cuda() if res is not None else None
    C = torch.zeros(M, n_out, dtype=torch.bfloat16, device="cuda")
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    rc = lib.vstar_op_gemm_fp8(None, P(dA), P(dW), P(dB), P(dR), P(C), M, N, K, epi, 0, None)
    assert rc == 0, lib.vstar_last_error(None)

    def fq(x):                                    # fake-quant: what the engine's operands decode to
        s = x.float().abs().amax(dim=1, keepdim=True) / 448.0
        s = torch.where(s > 0, s, torch.ones_like(s))
        return (x.float() * (1.0 / s)).to(torch.float8_e4m3fn).float(), s      # multiply by the fp32 reciprocal, like the kernel
    aq, sa = fq(A)
    wq, sw = fq(W[:N])
    ref = (aq @ wq.T) * sa * sw.T
    if bias is not None:
        ref = ref + bias[:N].float()
    if epi == 4:
        r = ref.view(M, N // 32, 2, 16)
        ref = (torch.nn.functional.silu(r[:, :, 0].bfloat16().float()).bfloat16().float() * r[:, :, 1].bfloat16().float()).reshape(M, n_out)
    if res is not None:
        ref = ref.bfloat16().float() + res.float()
    got = C.float().cpu()
    scale = float(ref.abs().max())
    err = float((got - ref).abs().max())
    assert err <= 1.2e-2 * scale, (err, scale)                # one bf16 ulp of the largest outputs
    assert float((got - ref).abs().mean()) <= 1.5e-3 * scale
    # and the quantisation itself is sane: W8A8 vs the unquantised product within a few percent
    full = A.float() @ W[:N].float().T
    if epi == 0:
        full = (full + bias[:N].float()).bfloat16().float() + res.float()
        assert float((got - full).norm() / full.norm()) < 0.06


# ---- RMSNorm folded into the LLaMA linears: row scale in the epilogue, next norm's statistics from the producing epilogue ----
def _gemm_norm(lib, a, w, res=None, epi=0, tile=0, row_scale=None, want_sumsq=False):
    M, K = a.shape
    N = w.shape[0]
    n_out = N // 2 if epi == _lib.EPI_SILU_MUL else N
    c = torch.full((M, n_out), float("nan"), dtype=torch.bfloat16, device=a.device)
    ss = torch.full((M, N // 64), float("nan"), dtype=torch.float32, device=a.device) if want_sumsq else None
    flag = {0: 0, 128: _lib.EPI_TILE128, 256: _lib.EPI_TILE256, _lib.TILE_4W: _lib.EPI_TILE4W}[tile]
    rc = lib.vstar_op_gemm_norm(None, P(a), K, P(_pad_w(w)), None, P(res), n_out if res is not None else 0, P(c), n_out, M, N, K,
                                epi | flag, P(row_scale), P(ss), N // 64)
    assert rc == 0, lib.vstar_last_error(None)
    if tile:
        assert lib.vstar_op_gemm_last_tile() == tile
    return c, ss


@pytest.mark.parametrize("M,N,K", [(1300, 4096, 256), (2048, 512, 1024), (1030, 1024, 128)])
def test_gemm_sumsq_partials_and_rstd(lib, cuda, M, N, K):
    """o_proj / down_proj form (x = A.W^T + residual): both kernels write bit-identical 64-column sums of squares of the stored
    values, rms_rstd from those partials == rms_rstd from the stored x itself == torch's fp32 statistics (to fp32 rounding)."""
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).bfloat16().to(cuda)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).bfloat16().to(cuda)
    res = (3 * torch.randn(M, N, generator=g)).bfloat16().to(cuda)
    c256, s256 = _gemm_norm(lib, a, w, res, tile=256, want_sumsq=True)
    c128, s128 = _gemm_norm(lib, a, w, res, tile=128, want_sumsq=True)
    plain = _gemm(lib, a, w, None, res, tile=256)
    assert torch.equal(c256, plain) and torch.equal(c128, plain)            # the statistics do not disturb the output
    assert torch.equal(s256, s128)                                          # same summation tree in both kernels
    ref = c256.float().view(M, N // 64, 64).pow(2).sum(-1)
    assert torch.allclose(s256, ref, rtol=1e-5, atol=0)
    eps = 1e-6
    r_part = torch.empty(M, dtype=torch.float32, device=cuda)
    r_rows = torch.empty(M, dtype=torch.float32, device=cuda)
    assert lib.vstar_op_rms_rstd(None, None, P(s256), N // 64, M, N, eps, P(r_part)) == 0
    assert lib.vstar_op_rms_rstd(None, P(c256), None, 0, M, N, eps, P(r_rows)) == 0
    torch.cuda.synchronize()
    assert torch.equal(r_part, r_rows)                                      # bit-identical: layer 0 (from x) and layers >= 1 (from partials)
    r_ref = torch.rsqrt(c256.float().pow(2).mean(-1) + eps)
    assert torch.allclose(r_part, r_ref, rtol=2e-6, atol=0)


@pytest.mark.parametrize("M,N,K,epi", [(1300, 768, 256, 0), (1100, 512, 256, 4), (2048, 4096, 512, 0)])
def test_gemm_row_scale_is_the_folded_rmsnorm(lib, cuda, M, N, K, epi):
    """Linear(RMSNorm(x)) == rstd[m] * (x . (W * norm_w)^T): the row scale acts on the fp32 accumulators (before SiLU*up / bias);
    128^2 and 256^2 kernels bit-identical; against torch fp32 at GEMM tolerance."""
    g = torch.Generator().manual_seed(M * 5 + N + K + epi)
    x = (2 * torch.randn(M, K, generator=g)).bfloat16().to(cuda)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).bfloat16()
    packed = _pack_gate_up(w[: N // 2], w[N // 2:]) if epi == _lib.EPI_SILU_MUL else w
    r = torch.empty(M, dtype=torch.float32, device=cuda)
    assert lib.vstar_op_rms_rstd(None, P(x), None, 0, M, K, 1e-6, P(r)) == 0
    c256, _ = _gemm_norm(lib, x, packed.to(cuda), epi=epi, tile=256, row_scale=r)
    c128, _ = _gemm_norm(lib, x, packed.to(cuda), epi=epi, tile=128, row_scale=r)
    assert torch.equal(c256, c128)
    acc = (x.float() @ w.to(cuda).float().T) * r[:, None]
    if epi == _lib.EPI_SILU_MUL:
        gf, uf = acc[:, : N // 2].bfloat16().float(), acc[:, N // 2:].bfloat16().float()
        ref = F.silu(gf).bfloat16().float() * uf
        assert ((c256.float() - ref).abs() <= ref.abs() * 2 ** -6 + 1e-2).all()
    else:
        assert _rel(c256.float(), acc) < 2 ** -7




@pytest.mark.parametrize("M,N,K,epi,use_bias,use_res", [
    (2048, 1024, 256, 0, False, False), (2048, 1024, 256, 0, True, True), (1536, 768, 768, 1, True, False),
    (2304, 512, 128, 2, True, True), (1280, 768, 256, 3, True, True), (2048, 2048, 512, 4, False, False),
    (1300, 1000, 256, 0, True, True),            # M and N tails: interior tiles direct, edge tiles through the LDS epilogue
])
def test_gemm256_direct_epilogue_equals_lds_epilogue(lib, cuda, monkeypatch, M, N, K, epi, use_bias, use_res):
    """Round 5: interior tiles of gemm256 finish in the accumulator registers (W rows DMA'd in a permuted order so that a lane's
    fragments are 16 contiguous output bytes).  VSTAR_GEMM_DEBUG=4 switches that off (every tile through the LDS transpose): both
    epilogues — and the 128^2 kernel — must agree bit for bit, statistics included."""
    g = torch.Generator().manual_seed(M + 7 * N + K + epi)
    a = torch.randn(M, K, generator=g).bfloat16().to(cuda)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).bfloat16()
    if epi == _lib.EPI_SILU_MUL:
        w = _pack_gate_up(w[: N // 2], w[N // 2:])
    w = w.to(cuda)
    n_out = N // 2 if epi == _lib.EPI_SILU_MUL else N
    bias = torch.randn(N, generator=g).bfloat16().to(cuda) if use_bias else None
    res = torch.randn(M, n_out, generator=g).bfloat16().to(cuda) if use_res else None
    monkeypatch.delenv("VSTAR_GEMM_DEBUG", raising=False)
    direct = _gemm(lib, a, w, bias, res, epi=epi, tile=256)
    c128 = _gemm(lib, a, w, bias, res, epi=epi, tile=128)
    monkeypatch.setenv("VSTAR_GEMM_DEBUG", "4")
    lds = _gemm(lib, a, w, bias, res, epi=epi, tile=256)
    assert not torch.isnan(direct.float()).any()
    assert torch.equal(direct, lds) and torch.equal(direct, c128)
    if epi == 0 and use_res and N % 64 == 0:
        monkeypatch.delenv("VSTAR_GEMM_DEBUG", raising=False)
        c_d, s_d = _gemm_norm(lib, a, w, res, tile=256, want_sumsq=True)
        monkeypatch.setenv("VSTAR_GEMM_DEBUG", "4")
        c_l, s_l = _gemm_norm(lib, a, w, res, tile=256, want_sumsq=True)
        assert torch.equal(c_d, c_l) and torch.equal(s_d, s_l)


@pytest.mark.parametrize("N,K", [(3072, 1024), (2304, 768)])
def test_ln_fold_zero_sum_rounding(lib, cuda, N, K):
    """The LayerNorm fold of the ViT linears: W' = round(W diag(g) - row mean) must SUM TO ZERO as stored, otherwise
    rstd * mean(x) * sum_k W'_k leaks into every output of rows with a large |mean| / std.  Nearest rounding leaves
    ~sqrt(K)/2 ulp; the zero-sum rounding of ln_fold_kernel moves a handful of elements to their other neighbour.  Checked here:
    (a) the stored row sums are <= 1e-3 of the nearest-rounding ones, (b) every element is within ONE 16-bit step of the exact
    centred value and all but a few within half a step, (c) the bias picks up W . b_ln, (d) a constant added to x does not move
    Linear(LN(x)) computed the folded way (x . W'^T, fp64 on the stored weights) by more than 1e-4 of the output scale, where
    nearest rounding moves it by ~|shift| * 2e-3."""
    g0 = torch.Generator().manual_seed(N + K)
    w = (torch.randn(N, K, generator=g0) / math.sqrt(K) * torch.exp(0.5 * torch.randn(N, K, generator=g0))).bfloat16()
    gam = (1.0 + 0.3 * torch.randn(K, generator=g0)).bfloat16()
    bln = (0.2 * torch.randn(K, generator=g0)).bfloat16()
    bias = (0.1 * torch.randn(N, generator=g0)).bfloat16()
    wd, bd = w.clone().to(cuda), bias.clone().to(cuda)
    assert lib.vstar_op_ln_fold(None, P(wd), P(bd), P(gam.to(cuda)), P(bln.to(cuda)), N, K) == 0, lib.vstar_last_error(None)
    torch.cuda.synchronize()
    exact = (w.float() * gam.float())
    exact = (exact - exact.mean(dim=1, keepdim=True))                        # fp32, like the kernel (same mean up to summation order)
    nearest = exact.bfloat16().double()
    got = wd.cpu().double()
    s_near, s_got = nearest.sum(1).abs(), got.sum(1).abs()
    # (what is left is a fraction of the smallest 16-bit step the row offers: ~1e-6 on rows whose nearest rounding leaves 2e-3)
    assert s_got.max() <= 5e-3 * s_near.mean() and s_got.mean() <= 5e-4 * s_near.mean(), (float(s_got.max()), float(s_near.mean()))
    step = (2.0 ** (torch.floor(torch.log2(exact.double().abs().clamp_min(1e-30))) - 7))       # bf16 spacing at each value
    dev = (got - exact.double()).abs()
    # (+ 1e-6: the kernel's fp32 row mean and this test's differ in summation order by ~1e-8, which is many 16-bit steps for the
    # handful of centred elements that land within 1e-7 of zero)
    assert (dev <= 1.02 * step + 1e-6).all()
    moved = (got != nearest).sum(1)
    assert moved.float().mean() < 40 and moved.max() < 120, (float(moved.float().mean()), int(moved.max()))
    want_b = (bias.double() + (w.double() * bln.double()).sum(1))
    assert torch.allclose(bd.cpu().double(), want_b, rtol=2 ** -7, atol=2e-3)
    # (d) shift invariance of the folded product
    x = torch.randn(64, K, generator=g0).double()
    shift = 8.0
    y0, y1 = x @ got.T, (x + shift) @ got.T
    n0, n1 = x @ nearest.T, (x + shift) @ nearest.T
    scale = float(y0.abs().mean())
    assert float((y1 - y0).abs().max()) <= 1e-4 * scale
    assert float((n1 - n0).abs().max()) > 1e-3 * scale           # the leak this removes
