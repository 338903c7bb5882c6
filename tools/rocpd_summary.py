#!/usr/bin/env python
"""Dumps the per-kernel summary (`top_kernels` view) of a rocprofv3 rocpd database as CSV.
usage: python tools/rocpd_summary.py gpurun_out/prof/bench_results.db > profiles/rNN_kernel_stats.csv"""
import csv
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vstar_amd.provenance import checked_hash  # noqa: E402
print(f"# kernel_source_hash {checked_hash()}")      # round 5: which kernel sources this profile belongs to
w = csv.writer(sys.stdout)
w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
for name, calls, tot, avg, pct in rows:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    if len(name) > 160:
        name = name[:157] + "..."
    w.writerow([name, calls, round(tot, 1), round(avg, 2), round(pct, 3)])
