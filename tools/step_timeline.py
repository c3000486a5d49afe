#!/usr/bin/env python3
"""Ordered kernel timeline of ONE step from a rocprofv3 results .db (rocpd sqlite).

    python tools/step_timeline.py results.db [marker-kernel-substring] > timeline.md

A step is the span between the last two launches of the marker kernel (default: pair_count, launched once per
geometry build).  Columns: start offset (us), duration (us), gap to the previous kernel's end (us), grid, name."""
import sqlite3
import sys


def main(path, marker="pair_count"):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    gx = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
    wx = "workgroup_x" if "workgroup_x" in cols else ("workgroup_size_x" if "workgroup_size_x" in cols else None)
    qx = next((c_ for c_ in ("stream_id", "queue_id", "stream", "queue") if c_ in cols), None)  # the HIP stream of a launch
    sel = "name, start, end" + (f", {gx}" if gx else ", 0") + (f", {wx}" if wx else ", 1") + (f", {qx}" if qx else ", 0")
    rows = c.execute(f"select {sel} from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    if len(marks) < 2:
        print("marker not found twice; columns:", cols)
        return
    a, b = marks[-2], marks[-1]
    t0 = rows[a][1]
    prev_end = t0
    print(f"# one step of `{path.split('/')[-1]}`: {b - a} launches, {(rows[b][1] - t0) / 1e3:.1f} us "
          f"(kernel time {sum(r[2] - r[1] for r in rows[a:b]) / 1e3:.1f} us)\n")
    print("| # | start us | dur us | gap us | workgroups | stream | kernel |")
    print("|---:|---:|---:|---:|---:|---:|---|")
    streams = {}
    for k, (n, s, e, g, w, q) in enumerate(rows[a:b]):
        n = n if len(n) < 100 else n[:97] + "..."
        sid = streams.setdefault(q, len(streams))
        print(f"| {k} | {(s - t0) / 1e3:.1f} | {(e - s) / 1e3:.1f} | {(s - prev_end) / 1e3:.1f} | {g // max(w, 1)} | {sid} | `{n}` |")
        prev_end = max(prev_end, e)


if __name__ == "__main__":
    main(*sys.argv[1:3])
