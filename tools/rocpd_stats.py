#!/usr/bin/env python3
"""Summarise a rocprofv3 results .db (rocpd sqlite) into the `--stats` style per-kernel table.

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db > profiles/r01_x_kernel_stats.md
"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print(f"# rocprofv3 --kernel-trace --stats summary of `{path.split('/')[-1]}`\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for n, k, s, a, mn, mx in rows:
        n = n if len(n) < 90 else n[:87] + "..."
        print(f"| `{n}` | {k} | {s / 1e6:.3f} | {a / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100 * s / tot:.2f} |")
    print(f"\ntotal kernel time {tot / 1e6:.3f} ms")


if __name__ == "__main__":
    main(sys.argv[1])
