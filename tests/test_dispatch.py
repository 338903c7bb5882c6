"""The GEMM dispatcher, pinned on the host (vstar_op_gemm_plan: the library takes every decision of gemm_lp for a device with
`cus` compute units and launches nothing — no GPU needed).  Every kernel / tile variant accumulates K in the same order, so a
change of this table never changes results (tests/test_ops_gpu.py::test_gemm128_equals_gemm256), only speed: the table is the
measured-best choice per shape of the path (profiles/r03_gemm_small_batch.txt, r03_gemm_bench_final.txt) and a dispatcher edit
that moves a shape must show up here.

code = 10 * tile + variant: tile 256 = the 256^2 kernel, 128 = the 128-row family, 384 = whole rounds of 256^2 tiles + one
round of 128^2 tiles for the ragged rest; variant 2 = double buffer (two workgroups per CU), 5 / 6 / 7 = loader-wave ring on the
128 x 128 / 128 x 64 / 128 x 256 tile."""
import pytest

from vstar_amd import _lib

S, NC, NO = 640, 577, 2305                       # rows per crop: LLaMA sequence, CLIP-L/14@336 tokens, OWL-ViT-B/16@768 tokens
# name: (rows per crop, N, K, epilogue, residual, fused RoPE where the 256^2 kernel runs)
SHAPES = {"qkv": (S, 12288, 4096, 0, 0, 1), "o": (S, 4096, 4096, 0, 1, 0), "gate_up": (S, 22016, 4096, 4, 0, 0),
          "down": (S, 4096, 11008, 0, 1, 0), "clip_qkv": (NC, 3072, 1024, 0, 0, 0), "clip_out": (NC, 1024, 1024, 0, 1, 0),
          "clip_fc1": (NC, 4096, 1024, 1, 0, 0), "clip_fc2": (NC, 1024, 4096, 0, 1, 0), "owl_qkv": (NO, 2304, 768, 0, 0, 0),
          "owl_out": (NO, 768, 768, 0, 1, 0), "owl_fc1": (NO, 3072, 768, 1, 0, 0), "owl_fc2": (NO, 768, 3072, 0, 1, 0)}
EXPECTED = {   # crops per step -> shape -> code, on the 256 CUs of an MI355X
    1: {'qkv': 1287, 'o': 1285, 'gate_up': 1287, 'down': 1285, 'clip_qkv': 1286, 'clip_out': 1286, 'clip_fc1': 1285, 'clip_fc2': 1286,
        'owl_qkv': 1282, 'owl_out': 1286, 'owl_fc1': 1282, 'owl_fc2': 1286},
    2: {'qkv': 25640, 'o': 1287, 'gate_up': 25640, 'down': 1287, 'clip_qkv': 1285, 'clip_out': 1286, 'clip_fc1': 1282, 'clip_fc2': 1286,
        'owl_qkv': 2560, 'owl_out': 1285, 'owl_fc1': 2560, 'owl_fc2': 1285},
    4: {'qkv': 25640, 'o': 25640, 'gate_up': 25640, 'down': 25640, 'clip_qkv': 1282, 'clip_out': 1285, 'clip_fc1': 2560, 'clip_fc2': 1285,
        'owl_qkv': 2560, 'owl_out': 1282, 'owl_fc1': 2560, 'owl_fc2': 1287},
    8: {'qkv': 25640, 'o': 3845, 'gate_up': 25640, 'down': 3845, 'clip_qkv': 2560, 'clip_out': 1282, 'clip_fc1': 3845, 'clip_fc2': 1287,
        'owl_qkv': 2560, 'owl_out': 2560, 'owl_fc1': 2560, 'owl_fc2': 2560},
    32: {'qkv': 25640, 'o': 25640, 'gate_up': 25640, 'down': 25640, 'clip_qkv': 2560, 'clip_out': 3845, 'clip_fc1': 2560, 'clip_fc2': 3845,
         'owl_qkv': 3845, 'owl_out': 2560, 'owl_fc1': 2560, 'owl_fc2': 2560},
}


def plan(lib, B, name, cus=256, flags=0):
    rows, N, K, epi, res, rope = SHAPES[name]
    M = B * rows
    return lib.vstar_op_gemm_plan(M, N, K, epi | flags, res, rope if M >= 1024 else 0, cus)


@pytest.mark.parametrize("B", sorted(EXPECTED))
def test_dispatch_table_of_the_path(B):
    lib = _lib.load()
    got = {name: plan(lib, B, name) for name in SHAPES}
    assert got == EXPECTED[B]


def test_dispatch_rules():
    lib = _lib.load()
    # the headline batch: every LLaMA linear on the 4-wave / AGPR 256^2 kernel (round 6; code 2564 = "256, 4 waves"), M % 256 == 0
    assert all(plan(lib, 32, n) == 25640 for n in ("qkv", "o", "gate_up", "down"))
    # the 128 x 256 loader-wave tile: long K and more 128^2 tiles than CUs only — never for the short-K ViT towers at one crop
    assert plan(lib, 1, "qkv") == plan(lib, 1, "gate_up") == 1287
    assert all(plan(lib, 1, n) % 10 != 7 for n in SHAPES if SHAPES[n][2] < 2048)
    # grids of at most one 128^2 tile per CU take a loader-wave ring (5: 128 x 128, 6: 128 x 64 below half-filled grids)
    assert plan(lib, 1, "o") == 1285 and plan(lib, 1, "clip_out") == 1286
    # the fused-RoPE epilogue exists only in the 256^2 kernel: asking for it outside that kernel's domain is an error, not a re-route
    assert lib.vstar_op_gemm_plan(640, 12288, 4096, 0, 0, 1, 256) < 0
    # rows that are no multiple of 256 (a prompt of another length): the LLaMA linears still take the 4-wave kernel (ragged last row
    # tile, K >= 4096), the short-K ViT shapes stay where they were
    assert all(lib.vstar_op_gemm_plan(32 * 623, N, K, e, r, ro, 256) == 25640
               for (N, K, e, r, ro) in ((12288, 4096, 0, 0, 1), (4096, 4096, 0, 1, 0), (22016, 4096, 4, 0, 0), (4096, 11008, 0, 1, 0)))
    # explicit per-call tile requests are honoured or refused, never silently changed
    assert plan(lib, 32, "o", flags=_lib.EPI_TILE128) // 10 == 128
    assert plan(lib, 32, "o", flags=_lib.EPI_TILE256) == 2560
    assert lib.vstar_op_gemm_plan(640, 4096, 4096, _lib.EPI_TILE256, 0, 0, 256) < 0
    # the decision depends on the device's CU count (MI300X: 304): same shapes, another table — and no launch, no GPU, either way
    assert plan(lib, 32, "o", cus=304) != plan(lib, 32, "o", cus=256)
    assert lib.vstar_op_gemm_plan(0, 16, 64, 0, 0, 0, 256) < 0
