#!/usr/bin/env python3
"""Summarise the roctx ranges ("tsd:<entry point>") of a rocprofv3 --marker-trace results .db (rocpd sqlite: the
`regions` view holds one row per range, category MARKER_CORE_RANGE_API, the range name in extdata.message).

    python tools/marker_summary.py results.db > profiles/rNN_markers.md"""
import collections
import json
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    acc = collections.OrderedDict()
    for ext, dur in c.execute("select extdata, duration from regions where category like 'MARKER%'"):
        try:
            name = json.loads(ext).get("message", "")
        except (TypeError, ValueError):
            continue
        if name.startswith("tsd:"):
            acc.setdefault(name, []).append(dur)
    print(f"# roctx ranges recorded by rocprofv3 --marker-trace in `{path.split('/')[-1]}`\n")
    if not acc:
        print("no `tsd:` ranges found")
        return
    print("| range (host side of the library call) | calls | total ms | avg us | min us | max us |")
    print("|---|---:|---:|---:|---:|---:|")
    for n, d in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        print(f"| `{n}` | {len(d)} | {sum(d) / 1e6:.3f} | {sum(d) / len(d) / 1e3:.2f} | {min(d) / 1e3:.2f} | {max(d) / 1e3:.2f} |")


if __name__ == "__main__":
    main(sys.argv[1])
