"""The generated K loop of the opt-in hand-scheduled GEMM (vstar_amd/csrc/gemm256a_loop.inc <- tools/gen_gemm256a_asm.py): the
committed include is what the generator writes, and the instruction stream has the structure gemm256a.hip relies on (its results
are checked bit for bit against gemm256 on the GPU: tests/test_ops_gpu.py::test_gemm256a_bit_identical_to_gemm256)."""
import importlib.util
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gen(tmp_path):
    spec = importlib.util.spec_from_file_location("gen_gemm256a_asm", os.path.join(ROOT, "tools", "gen_gemm256a_asm.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = str(tmp_path / "loop.inc")
    return mod.main(out), out


def test_committed_include_is_the_generators_output(tmp_path):
    _, out = _gen(tmp_path)
    with open(out) as f, open(os.path.join(ROOT, "vstar_amd", "csrc", "gemm256a_loop.inc")) as g:
        assert f.read() == g.read()


def test_loop_structure(tmp_path):
    L, _ = _gen(tmp_path)
    i0, i1 = L.index(".Lg256a_loop_%=:"), L.index(".Lg256a_tail_%=:")
    loop, tail = L[i0:i1], L[i1:]
    # two K-tiles per iteration, 128 MFMAs each (8 x 8 fragments x 2 k-halves); the peeled last pair likewise
    assert sum("v_mfma" in x for x in loop) == 256 and sum("v_mfma" in x for x in tail) == 256
    # every accumulator fragment a[4j : 4j+3] is written exactly once per k-half: 4 times per loop iteration, in place (C = D)
    acc = [re.search(r"v_mfma_f32_16x16x32_bf16 a\[(\d+):(\d+)\], .*a\[(\d+):(\d+)\]$", x).groups() for x in loop if "v_mfma" in x]
    assert all(a == c and b == d and int(b) == int(a) + 3 and int(a) % 4 == 0 for a, b, c, d in acc)
    counts = {}
    for a, *_ in acc:
        counts[a] = counts.get(a, 0) + 1
    assert len(counts) == 64 and set(counts.values()) == {4}
    # per K-tile and wave: 32 fragment reads, 16 LDS-DMA pieces (each with its own m0), one barrier; the tail issues no DMA
    assert sum(x.startswith("ds_read_b128") for x in loop) == 64
    assert sum("global_load_lds_dwordx4" in x for x in loop) == 32 and sum(x.startswith("s_add_i32 m0") for x in loop) == 32
    assert sum(x == "s_barrier" for x in loop) == 2
    assert not any("global_load_lds" in x for x in tail)
    # the only waits on the vector-memory counter sit in front of the barriers; with the L2 prefetch duty they leave exactly the two
    # prefetch loads (the youngest requests) in flight
    vm = [x for x in loop if "vmcnt" in x]
    assert vm == ["s_waitcnt vmcnt(2) lgkmcnt(0)"] * 2
    assert sum(x.startswith("global_load_dword ") for x in loop) == 4
    # operand fragments live in v0..v127 (two sets), addresses above; nothing in the loop writes an address register except the
    # k advance of the DMA / prefetch offsets
    for x in loop:
        m = re.match(r"ds_read_b128 v\[(\d+):(\d+)\]", x)
        if m:
            assert int(m.group(2)) < 128
