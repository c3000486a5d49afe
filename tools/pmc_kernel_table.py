#!/usr/bin/env python3
"""Per-kernel averages of every collected PMC counter plus the average duration, from a rocprofv3 rocpd .db
(views `counters_collection` and `kernels`).   python tools/pmc_kernel_table.py x_results.db [name-filter]"""
import sqlite3
import sys

db, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
c = sqlite3.connect(db)
dur = {r[0]: (r[1], r[2]) for r in c.execute("select name, avg(duration), count(*) from kernels group by name")}
rows = c.execute("select kernel_name, counter_name, avg(value) from counters_collection group by kernel_name, counter_name")
out = {}
for k, n, v in rows:
    out.setdefault(k, {})[n] = v
for k, d in out.items():
    if "tsd::" not in k or flt not in k:
        continue
    ns, cnt = dur.get(k, (0.0, 0))
    print(k.replace("void ", "").replace("tsd::", "").split("(")[0], f"  avg {ns / 1e3:.2f} us over {cnt} launches")
    for n, v in sorted(d.items()):
        extra = ""
        if n == "GRBM_GUI_ACTIVE" and ns:
            extra = f"   -> effective clock {v / ns:.3f} GHz"
        print(f"   {n:28s} {v:16.0f}{extra}")
