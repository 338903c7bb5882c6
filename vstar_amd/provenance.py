"""Provenance stamp for committed evidence: a hash of the KERNEL SOURCES (vstar_amd/csrc/*.hip, *.hpp,
build.sh).  tools/pmc_summary.py / tools/rocpd_summary.py write it into the summaries they produce on the GPU box (where .git does
not exist — the snapshot ships sources only), and bench.py quotes a committed PMC figure only when that file's stamp equals the
hash of the sources it is running: host-only commits do not stale the evidence, a kernel edit does."""
from __future__ import annotations

import glob
import hashlib
import os

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")


def kernel_source_hash() -> str:
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(_CSRC, "*.hip")) + glob.glob(os.path.join(_CSRC, "*.hpp")) + [os.path.join(_CSRC, "build.sh")]):
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]
