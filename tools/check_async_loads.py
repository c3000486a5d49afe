#!/usr/bin/env python3
"""Static check of the inline-asm global loads in the compiled kernels (gfx950 ISA text, `hipcc -S`).

The tile GEMMs feed their B operand through a ring of `global_load_dwordx4` issued by inline asm and consumed behind
counted `s_waitcnt vmcnt(N)` statements (csrc/common.hpp, csrc/split16.hpp); the gathers do the same with one wait per
batch.  For the compiler an asm output is written AT the statement, so nothing in the language stops it from copying,
spilling or re-using such a register between the load and its wait -- it does so only under register pressure (a
`__launch_bounds__` occupancy cap), silently, and the kernel then computes with stale fragments.  This script proves
the absence of that for the BINARY: it walks every kernel in text order, keeps the hardware's vmcnt FIFO (every vector
memory instruction counts; loads return in order, stores in order, the two classes not with respect to each other),
and reports any instruction that names a destination register of an asm-issued load that a preceding wait has not
covered.  The walk is linear (branches are not followed): conservative inside straight-line tile code, which is where
the rings live.

    python tools/check_async_loads.py file.s [...]        exit code 1 on a violation
"""
import re
import sys

VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
VMEM = re.compile(r"^\s*(global_|buffer_|scratch_|flat_)(load|store|atomic)")
WAIT = re.compile(r"s_waitcnt\b.*?vmcnt\((\d+)\)")
KERNEL = re.compile(r"^(_Z\w+):\s*(;.*)?$")


def regs_of(text):
    out = set()
    for m in VREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def check_kernel(name, lines):
    """`maybe`: the vector memory operations that MAY still be outstanding, in issue order; `out_max`: an upper bound
    of the hardware counter.  A wait vmcnt(n) leaves at most n outstanding, so at least len(maybe) - n of them are back:
    loads return in order among loads, stores among stores, so the oldest (that many - number of the other class)
    entries of each class are certainly back."""
    maybe = []     # [is_load, frozenset(dest regs) for an asm-issued load or None, line]
    out_max = 0
    hot = {}       # register -> line of the asm load that has it in flight
    in_asm = False
    bad = []

    def retire(entries):
        for e in entries:
            if e[1]:
                for r in e[1]:
                    hot.pop(r, None)

    for no, raw in lines:
        st = raw.strip()
        if st.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if st.startswith(";;#ASMEND"):
            in_asm = False
            continue
        line = raw.split(";", 1)[0].strip()
        if not line or line.startswith(".") or line.endswith(":"):
            continue
        w = WAIT.search(line)
        if w:
            n = int(w.group(1))
            out_max = min(out_max, n)
            must = len(maybe) - out_max
            if must > 0:
                loads = [e for e in maybe if e[0]]
                stores = [e for e in maybe if not e[0]]
                gl = max(0, must - len(stores))   # loads certainly back
                gs = max(0, must - len(loads))    # stores certainly back
                gone = loads[:gl] + stores[:gs]
                retire(gone)
                ids = {id(e) for e in gone}
                maybe = [e for e in maybe if id(e) not in ids]
            continue
        used = regs_of(line)
        clash = used & hot.keys()
        if clash:
            bad.append((no, raw.strip(), sorted(clash), sorted({hot[r] for r in clash})))
        if VMEM.match(line) is not None:
            is_load = "_load" in line.split()[0]
            dest = None
            if in_asm and is_load:
                first = line.split(None, 1)[1].split(",")[0]
                dest = frozenset(regs_of(first))
                for r in dest:
                    hot[r] = no
            maybe.append([is_load, dest, no])
            out_max += 1
    return bad


def main(paths):
    total = 0
    kernels = 0
    for path in paths:
        cur, body = None, []
        items = []
        with open(path) as f:
            for no, raw in enumerate(f, 1):
                m = KERNEL.match(raw)
                if m:
                    if cur:
                        items.append((cur, body))
                    cur, body = m.group(1), []
                elif raw.lstrip().startswith(".end_amdhsa_kernel") or raw.lstrip().startswith(".section"):
                    if cur:
                        items.append((cur, body))
                    cur, body = None, []
                elif cur:
                    body.append((no, raw.rstrip("\n")))
        if cur:
            items.append((cur, body))
        for name, body in items:
            if not any(";;#ASMSTART" in l for _, l in body):
                continue
            kernels += 1
            bad = check_kernel(name, body)
            for no, text, regs, src in bad[:8]:
                print(f"{path}:{no}: {name[:70]}: `{text}` touches v{regs} while the asm load(s) of line {src} are in flight")
            if len(bad) > 8:
                print(f"{path}: {name[:70]}: ... {len(bad) - 8} more")
            total += len(bad)
    print(f"check_async_loads: {kernels} kernels with inline asm, {total} violation(s)")
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
