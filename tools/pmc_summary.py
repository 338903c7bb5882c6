"""Per-kernel PMC summary from three separate rocprofv3 --pmc passes of the same command (MI355X_MICROARCH.md, HBM section:
one counter set per pass, FETCH_SIZE is in KiB and under-reports by 2x on gfx950 (128-byte requests counted as 64), WRITE_SIZE
in KiB; GRBM_GUI_ACTIVE is summed over the 8 XCDs, SQ_VALU_MFMA_BUSY_CYCLES over all SIMDs).

  python tools/pmc_summary.py <fetch.db> <write.db> <mfma.db> > profiles/rNN_pmc.json

  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out/f -o pmc -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d out/w -o pmc -- ...
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace -d out/m -o pmc -- ...
"""
import json
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*", "", name)
    return re.sub(r"\b(vs_bf16|vs_f16)::", "", name)


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    acc, n = defaultdict(float), defaultdict(int)
    for k, v in db.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
        acc[short(k)] += v
        n[short(k)] += 1
    return acc, n


def main():
    fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
    write, _ = per_kernel(sys.argv[2], "WRITE_SIZE")
    mfma, _ = per_kernel(sys.argv[3], "SQ_VALU_MFMA_BUSY_CYCLES")
    gui, _ = per_kernel(sys.argv[3], "GRBM_GUI_ACTIVE")
    out = []
    for k in sorted(fetch, key=lambda k: -fetch[k]):
        if fetch[k] * 2 * 1024 < 1e9 and mfma.get(k, 0) == 0:
            continue
        rec = {"kernel": k, "launches": nf[k], "fetch_GB_corrected": round(fetch[k] * 2 * 1024 / 1e9, 2),
               "write_GB": round(write.get(k, 0) * 1024 / 1e9, 2)}
        if gui.get(k):
            # busy cycles are summed over 4 SIMDs x 256 CUs; GUI_ACTIVE over 8 XCDs -> per-XCD active cycles x 1024 SIMDs
            rec.update(mfma_busy_cycles=mfma.get(k, 0), grbm_gui_active_sum_over_8_xcd=gui[k],
                       mfma_busy_frac=round(mfma.get(k, 0) / (gui[k] / 8 * 1024), 3))
        out.append(rec)
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from vstar_amd.provenance import checked_hash
    # stamped with the hash compiled into the library the profiled command LOADED (round 6; checked_hash() refuses to stamp when the
    # tree's sources are not the ones that library was built from); bench.py quotes only a file carrying its own library's hash
    json.dump({"kernel_source_hash": checked_hash(), "kernels": out}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
