"""Per-kernel sums of every counter found in one or more rocprofv3 --pmc result databases (one counter set per pass).
usage: python tools/pmc_dump.py out/a/pmc_results.db [out/b/pmc_results.db ...] > profiles/rNN_sq_counters.json"""
import json
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\((vs_bf16|vs_f16|unsigned|int|float|void|const|long).*", "", name)
    return re.sub(r"\b(vs_bf16|vs_f16)::", "", name)


acc = defaultdict(lambda: defaultdict(float))
launches = defaultdict(int)
have = set()                     # counters already taken from an earlier pass (a counter listed in two passes is not summed twice)
for path in sys.argv[1:]:
    db = sqlite3.connect(path)
    seen = defaultdict(set)
    this = set()
    for k, c, v, d in db.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection"):
        if c in have:
            continue
        this.add(c)
        acc[short(k)][c] += v
        seen[short(k)].add(d)
    have |= this
    for k, s in seen.items():
        launches[k] = max(launches[k], len(s))
out = []
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", acc[k].get("GRBM_GUI_ACTIVE", 0))):
    rec = {"kernel": k, "launches": launches[k]}
    rec.update({c: v for c, v in sorted(acc[k].items())})
    w = rec.get("SQ_WAVE_CYCLES")
    if w:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
                  "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC"):
            if c in rec:
                rec[c + "/WAVE_CYCLES"] = round(rec[c] / w, 4)
    out.append(rec)
json.dump(out, sys.stdout, indent=1)
