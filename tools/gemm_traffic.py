#!/usr/bin/env python
"""L2 <-> fabric traffic of the three GEMM kernels on ONE shape, for rocprofv3 --pmc (round 6).

Two modes:
  run:      python tools/gemm_traffic.py run SHAPE [launches]      -- launches hipBLASLt, gemm256 and gemm4w `launches` times each
            (under `rocprofv3 --pmc FETCH_SIZE --kernel-trace -d DIR -o pmc -- ...`, one counter set per pass)
  summary:  python tools/gemm_traffic.py summary NAME=pmc_results.db ...  -- per kernel and counter: mean value per launch
FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE is doubled here (gfx950 tallies 128-byte requests at 64, MI355X_MICROARCH.md).
hipBLASLt appears as a measurement yardstick only — nothing under vstar_amd/ links it."""
import ctypes
import os
import re
import sqlite3
import sys
from collections import defaultdict

SHAPES = {"qkv": (20480, 12288, 4096, 0), "o": (20480, 4096, 4096, 0), "gate_up": (20480, 22016, 4096, 0),
          "gate_up_silu": (20480, 22016, 4096, 4), "down": (20480, 4096, 11008, 0)}


def run(shape, launches):
    import torch
    import torch.nn.functional as F
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from vstar_amd import _lib
    lib = _lib.load()
    M, N, K, epi = SHAPES[shape]
    dev = torch.device("cuda:0")
    a = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    n_out = N // 2 if epi == 4 else N
    c = torch.empty(M, n_out, device=dev, dtype=torch.bfloat16)
    P = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    torch.backends.cuda.preferred_blas_library("hipblaslt")
    for _ in range(launches):
        if not epi:
            F.linear(a, w)
        assert lib.vstar_op_gemm(None, P(a), K, P(w), None, None, n_out, P(c), n_out, 0, M, N, K, epi | _lib.EPI_TILE256) == 0
        assert lib.vstar_op_gemm(None, P(a), K, P(w), None, None, n_out, P(c), n_out, 0, M, N, K, epi | _lib.EPI_TILE4W) == 0
    torch.cuda.synchronize()


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*", "", name)
    name = re.sub(r"\b(vs_bf16|vs_f16)::", "", name)
    return name[:60]


def summary(args):
    rows = defaultdict(dict)
    for a in args:
        tag, path = a.split("=", 1)
        db = sqlite3.connect(path)
        acc, n = defaultdict(float), defaultdict(int)
        for k, cn, v in db.execute("select kernel_name, counter_name, value from counters_collection"):
            acc[(short(k), cn)] += v
            n[(short(k), cn)] += 1
        for (k, cn), v in acc.items():
            per = v / n[(k, cn)]
            if cn == "FETCH_SIZE":
                rows[(tag, k)]["fetch_GB"] = round(per * 2 * 1024 / 1e9, 3)
            elif cn == "WRITE_SIZE":
                rows[(tag, k)]["write_GB"] = round(per * 1024 / 1e9, 3)
            else:
                rows[(tag, k)][cn] = round(per)
            rows[(tag, k)]["launches"] = n[(k, cn)]
    for (tag, k), r in sorted(rows.items()):
        if r.get("fetch_GB", 1) < 0.05 and "gemm" not in k and "Cijk" not in k:
            continue
        print(f"{tag:<14s} {k:<62s} " + "  ".join(f"{a}={b}" for a, b in sorted(r.items())))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 3)
    else:
        summary(sys.argv[2:])
