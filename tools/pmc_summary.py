#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE collected separately:
they do not fit one pass on gfx950, MI355X_MICROARCH.md 'rocprofv3 PMC slots').

    python tools/pmc_summary.py gpurun_out/pmc_fetch/f_results.db gpurun_out/pmc_write/w_results.db \
        profiles/r01_pmc_traffic

Units/corrections (MI355X_MICROARCH.md 'HBM'): both counters are in KiB-sized units of 1024 B here (the
WRITE_SIZE of filter_gen equals its 7 x 13 037 x 1 KiB output to 0.0 %); on gfx950 FETCH_SIZE reports
exactly HALF of the bytes of a wide coalesced read, so the read side is doubled ("fetch_bytes_corrected").
Infinity-Cache hits are counted, so this is memory-side traffic, an upper bound on DRAM traffic.
"""
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def stamp():
    """what the counters describe: hash of the kernel sources (bench.py refuses a file whose stamp is not the tree's) and
    the commit they were taken at (.git_head, written by the caller before `gpurun`: the GPU box has no .git)"""
    from bench import ROOT, kernel_source_sha
    head = os.environ.get("TSDIFF_HEAD")
    if not head:
        try:
            head = open(os.path.join(ROOT, ".git_head")).read().strip()
        except OSError:
            head = "unknown"
    return {"kernel_source_sha": kernel_source_sha(), "head": head}


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, count(*), avg(value), min(value), max(value) from counters_collection "
                     "where counter_name=? group by kernel_name", (counter,)).fetchall()
    return {r[0]: {"launches": r[1], "avg": r[2], "min": r[3], "max": r[4]} for r in rows}


def short(name):
    n = name.replace("void ", "").replace("tsd::", "")
    return n.split("(")[0]


def main(fetch_db, write_db, out_prefix):
    f = per_kernel(fetch_db, "FETCH_SIZE")
    w = per_kernel(write_db, "WRITE_SIZE")
    out = {}
    for k in sorted(set(f) | set(w)):
        if "tsd::" not in k:
            continue
        fe = f.get(k, {"avg": 0.0, "launches": 0})
        wr = w.get(k, {"avg": 0.0, "launches": 0})
        out[short(k)] = {
            "launches_profiled": fe["launches"],
            "fetch_bytes_raw": fe["avg"] * 1024.0,
            "fetch_bytes_corrected": fe["avg"] * 1024.0 * 2.0,
            "write_bytes": wr["avg"] * 1024.0,
            "hbm_bytes_per_launch": fe["avg"] * 2048.0 + wr["avg"] * 1024.0,
        }
    with open(out_prefix + ".json", "w") as fh:
        json.dump({**stamp(), "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of "
                             "`python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline`",
                   "correction": "FETCH_SIZE x2 (gfx950 wide-read under-count), KiB units", "kernels": out}, fh, indent=1)
    with open(out_prefix + ".md", "w") as fh:
        fh.write("# HBM-side traffic per launch (rocprofv3 PMC, FETCH_SIZE x2 corrected + WRITE_SIZE)\n\n")
        fh.write("| kernel | launches | fetch MB (corrected) | write MB | total MB |\n|---|---:|---:|---:|---:|\n")
        for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"]):
            fh.write(f"| `{k}` | {v['launches_profiled']} | {v['fetch_bytes_corrected'] / 1e6:.2f} | "
                     f"{v['write_bytes'] / 1e6:.2f} | {v['hbm_bytes_per_launch'] / 1e6:.2f} |\n")
    print(open(out_prefix + ".md").read())


if __name__ == "__main__":
    main(*sys.argv[1:4])
