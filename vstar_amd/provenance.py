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


def library_source_hash() -> str:
    """The hash compiled into the LOADED libvstar_hip.so (build.sh -> vstar_build_source_hash): which sources the binary that is
    actually running was built from.  Evidence is stamped with THIS value; `kernel_source_hash()` of the tree must equal it or the
    build is stale (a source hash taken at summary time does not identify the profiled binary)."""
    from . import _lib
    return _lib.load().vstar_build_source_hash().decode()


def checked_hash() -> str:
    """library_source_hash(), after checking that the tree's kernel sources are the ones the loaded library was built from."""
    lib_h, src_h = library_source_hash(), kernel_source_hash()
    if lib_h != src_h:
        raise RuntimeError(f"libvstar_hip.so was built from kernel sources {lib_h}, the tree holds {src_h}: rebuild "
                           "(__graft_entry__.build()) before producing or quoting evidence")
    return lib_h
