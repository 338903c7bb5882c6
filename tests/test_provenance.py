"""Provenance of committed evidence: summaries are stamped with a hash of the KERNEL sources, and bench.py only
quotes a PMC figure measured on the kernels it is running.  CPU-only."""
import json
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_kernel_source_hash_is_stable_and_tracks_kernel_edits(tmp_path, monkeypatch):
    from vstar_amd import provenance
    h = provenance.kernel_source_hash()
    assert len(h) == 16 and int(h, 16) >= 0 and h == provenance.kernel_source_hash()
    # a copy of csrc with one kernel byte changed hashes differently; a host-only edit elsewhere does not enter at all
    dst = tmp_path / "csrc"
    shutil.copytree(os.path.join(ROOT, "vstar_amd", "csrc"), dst, ignore=shutil.ignore_patterns("build"))
    monkeypatch.setattr(provenance, "_CSRC", str(dst))
    assert provenance.kernel_source_hash() == h
    with open(dst / "gemm256.hip", "a") as f:
        f.write("\n// edited\n")
    assert provenance.kernel_source_hash() != h


def test_pmc_summaries_of_this_round_carry_a_stamp():
    """Every r05+ PMC summary under profiles/ is the stamped form ({"kernel_source_hash", "kernels"}); older rounds' files are plain
    lists and are never quoted by bench.py any more."""
    import glob
    for path in glob.glob(os.path.join(ROOT, "profiles", "r0[5-9]_pmc*.json")):
        d = json.load(open(path))
        assert isinstance(d, dict) and len(d["kernel_source_hash"]) == 16 and isinstance(d["kernels"], list), path


def test_power_sampler_parses_rocm_smi_json(tmp_path, monkeypatch):
    """bench.py::_sample_power with a stand-in rocm-smi on PATH: mean power / shader clock over the samples taken while `step` runs."""
    fake = tmp_path / "rocm-smi"
    fake.write_text("#!/bin/sh\necho '{\"card0\": {\"Current Socket Graphics Package Power (W)\": \"1398.0\", "
                    "\"sclk clock speed:\": \"(1737Mhz)\", \"sclk clock level:\": \"1\", \"mclk clock speed:\": \"(2000Mhz)\"}}'\n")
    fake.chmod(0o755)
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "class _T:\n    @staticmethod\n    def synchronize(): pass\n"
            "bench.torch.cuda.synchronize = lambda: None\n"
            "print(__import__('json').dumps(bench._sample_power(lambda: time.sleep(0.05), seconds=0.9)))\n") % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120,
                         env=dict(os.environ, PATH=str(tmp_path) + os.pathsep + os.environ["PATH"]))
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["power_w"] == 1398.0 and d["sclk_mhz"] == 1737.0 and d["samples"] >= 2


def test_library_carries_the_hash_of_the_sources_it_was_built_from(lib):
    """the stamp identifies the BINARY — build.sh compiles kernel_source_hash() into the library; a tree whose kernel
    sources differ from the loaded library's is refused by checked_hash()."""
    from vstar_amd import provenance
    assert provenance.library_source_hash() == provenance.kernel_source_hash() == provenance.checked_hash()


@pytest.mark.parametrize("dtype_flag", [[], ["--f16"]])
def test_gemm4w_accumulators_are_never_overwritten_before_they_are_read(dtype_flag):
    """gemm4w.hip keeps its accumulators in AGPRs behind the compiler's back (clobbers of the K-loop statement, read back by asm
    statements in the epilogue).  tools/check_gemm4w_agpr.py walks the compiler's own assembly of all eight kernels and proves that
    nothing writes an AGPR before the epilogue has consumed it (round 6: the register allocator once parked the W piece offsets in
    a2..a9).  Needs hipcc (cross-compiles without a GPU); ~40 s."""
    import shutil as _sh
    if not (_sh.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("no hipcc")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_gemm4w_agpr.py")] + dtype_flag, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok ") == (12 if dtype_flag else 22)
