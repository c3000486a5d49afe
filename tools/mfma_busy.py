#!/usr/bin/env python3
"""MFMA-busy fraction per kernel from rocprofv3 SQ-counter passes (rocpd .db files of tools/profile_round.sh):
    frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x average launch duration x 2.4 GHz)
python tools/mfma_busy.py label=x_results.db ... > profiles/rNN_mfma_busy.json   (bench.py reads it)"""
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import stamp

out = dict(stamp())
for arg in sys.argv[1:]:
    label, db = arg.split("=", 1)
    try:
        c = sqlite3.connect(db)
        dur = {r[0]: (r[1], r[2]) for r in c.execute("select name, avg(duration), count(*) from kernels group by name")}
        rows = c.execute("select kernel_name, counter_name, avg(value) from counters_collection group by kernel_name, counter_name").fetchall()
    except Exception as e:  # a pass that did not run
        out[label] = {"error": str(e)}
        continue
    cnt = {}
    for k, n, v in rows:
        cnt.setdefault(k, {})[n] = v
    res = {}
    for k, d in cnt.items():
        if "tsd::" not in k or "SQ_VALU_MFMA_BUSY_CYCLES" not in d:
            continue
        ns, n = dur.get(k, (0.0, 0))
        if ns <= 0:
            continue
        name = k.replace("void ", "").replace("tsd::", "").split("(")[0]
        res[name] = {"avg_us": round(ns / 1e3, 2), "launches": n,
                     "mfma_busy": round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * ns * 2.4), 4)}
    out[label] = res
print(json.dumps(out, indent=1, sort_keys=True))
