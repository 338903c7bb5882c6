#!/usr/bin/env python
"""Proves, on the compiler's own assembly, that nothing but gemm4w's hand-written statements touches the AGPRs while they hold the
accumulators (round 6).  gemm4w.hip keeps a 128 x 128 fp32 tile per wave in a0..a255 across the K-loop asm statement and reads it back
with v_accvgpr_read statements in the epilogue; the register allocator does not know that and, under pressure, parks VGPR values in
AGPRs.  Every epilogue read statement clobbers the whole AGPR file, so no compiler value can live in an AGPR ACROSS a read; what is
left is short-term parking between two reads, and that is harmless exactly when the AGPR has already been consumed.  For every gemm4w
kernel this script walks the assembly from the LAST MFMA of the loop text to the last AGPR access and checks: each of a0..a255 is read
(its accumulator consumed) before anything writes it, all 256 are consumed, and no MFMA / v_accvgpr_mov appears.
usage: python tools/check_gemm4w_agpr.py [--f16]   (exit code 1 and a report on violation)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vstar_amd", "csrc")
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form -mllvm -amdgpu-spill-vgpr-to-agpr=0"


def main():
    extra = ["-DVSTAR_LP_F16"] if "--f16" in sys.argv else []
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "gemm4w.s")
        cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + FLAGS.split() + extra + ["-S", "--cuda-device-only", "gemm4w.hip", "-o", out]
        subprocess.check_call(cmd, cwd=CSRC, stderr=subprocess.DEVNULL)
        lines = open(out).read().split("\n")
    starts = [i for i, x in enumerate(lines) if re.match(r"^_ZN.*gemm4w_kernel.*:", x)]
    want = 12 if extra else 22           # {NONE, QUICK_GELU, RELU, SILU_MUL} x {plain, prefetch} x {whole tiles, ragged last row tile} (+ eight W8A8 and four block-scaled W8A8 kernels in the bf16 build)
    assert len(starts) == want, f"expected {want} gemm4w kernels, found {len(starts)}"
    bad = 0
    for si, s in enumerate(starts):
        seg = lines[s:starts[si + 1] if si + 1 < len(starts) else len(lines)]
        mf = [i for i, x in enumerate(seg) if "v_mfma" in x]
        rd = [i for i, x in enumerate(seg) if "v_accvgpr_read" in x]
        window = seg[mf[-1] + 1:]
        consumed, parked, offenders = set(), set(), []
        for x in window:
            x = x.strip()
            m = re.match(r"v_accvgpr_read_b32 v\d+, a\[?(\d+)(?:\+(\d+))?\]?", x)      # (the epilogue's statements print a[B+i])
            if m:
                n = int(m.group(1)) + int(m.group(2) or 0)
                if n in parked:
                    parked.discard(n)            # the compiler reading back what it parked
                else:
                    consumed.add(n)              # an epilogue read statement: the accumulator is consumed
                continue
            m = re.match(r"v_accvgpr_write_b32 a\[?(\d+)\]?,", x)
            if m:
                n = int(m.group(1))
                if n not in consumed:
                    offenders.append(x)          # overwrites an accumulator the epilogue has not read yet
                parked.add(n)
                continue
            if re.search(r"v_accvgpr_mov|v_mfma", x):
                offenders.append(x)
        name = seg[0].split(":")[0]
        ok = not offenders and len(consumed) == 256
        print(("ok   " if ok else "FAIL ") + name[-44:], "accumulators consumed", len(consumed), "violations", len(offenders), offenders[:4])
        bad += not ok
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
