#!/usr/bin/env python3
"""Static check of the inline-asm global loads in the compiled kernels (gfx950 ISA text, `hipcc -S`).

The tile GEMMs feed their B operand through a ring of `global_load_dwordx4` issued by inline asm and consumed behind
counted `s_waitcnt vmcnt(N)` statements (csrc/common.hpp, csrc/split16.hpp); the gathers do the same with one wait per
batch.  For the compiler an asm output is written AT the statement, so nothing in the language stops it from copying,
spilling or re-using such a register between the load and its wait -- it does so only under register pressure (a
`__launch_bounds__` occupancy cap), silently, and the kernel then computes with stale fragments.  This script proves
the absence of that for the BINARY: it walks every kernel in text order with the hardware's vmcnt rule (loads return
in order among loads; stores only add to the outstanding count), carries the set of in-flight registers along forward
branches to their labels, and reports any instruction that names a destination register of an asm-issued load that a
preceding wait has not covered on some path.

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


LABEL = re.compile(r"^(\.LBB\d+_\d+):")
BRANCH = re.compile(r"^\s*(s_branch|s_cbranch_\w+)\s+(\.LBB\d+_\d+)")


def merge(a, b):
    """two states reaching one label: a register is in flight if it is on either path; `younger` (the loads issued
    after it -- what a counted wait is measured against) is the smaller of the two"""
    if a is None:
        return None if b is None else dict(b)
    if b is None:
        return dict(a)
    out = dict(a)
    for r, (line, y) in b.items():
        out[r] = (line, min(y, out[r][1])) if r in out else (line, y)
    return out


def check_kernel(name, lines):
    """`hot`: register -> (line of the asm load that has it in flight, number of vector memory LOADS issued after that
    load).  Loads return in order among loads, so behind `s_waitcnt vmcnt(n)` -- at most n operations outstanding --
    a load with at least n younger loads is back (were it outstanding, so were they: more than n); stores only add to
    the outstanding count, so they never make a load look complete.
    Control flow: forward branches carry the state to their label, where it is merged with the fall-through state
    (a path that skips a wait keeps its loads in flight; a block reached only by a branch from before the loads has
    none); code behind an unconditional branch is unreachable until the next label.  `s_cbranch_execz/execnz` only skip
    lane-masked code and carry nothing: the walk goes through the skipped code (its conditions are correlated with the
    later ones in ways a merge would lose -- a wait inside `if (last)` and the `if (last && tid == 0)` behind it).
    Backward branches are not followed: the tile code never carries an asm load over a loop edge, and the loop body is checked on its first pass."""
    hot = {}
    pending = {}   # label -> state carried by forward branches
    seen = set()
    in_asm = False
    bad = []
    for no, raw in lines:
        st = raw.strip()
        if st.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if st.startswith(";;#ASMEND"):
            in_asm = False
            continue
        lab = LABEL.match(st)
        if lab:
            seen.add(lab.group(1))
            hot = merge(hot, pending.pop(lab.group(1), None))
            if hot is None:
                hot = {}   # (reached by a backward branch only: a loop header behind an unconditional branch)
            continue
        line = raw.split(";", 1)[0].strip()
        if not line or line.startswith(".") or line.endswith(":"):
            continue
        if hot is None:
            continue       # unreachable
        br = BRANCH.match(line)
        if br and br.group(1) in ("s_cbranch_execz", "s_cbranch_execnz"):
            continue       # a skip over lane-masked code: walking through that code is the conservative reading
        if br:
            if br.group(2) not in seen:
                pending[br.group(2)] = merge(pending.get(br.group(2)), hot)
            if br.group(1) == "s_branch":
                hot = None
            continue
        if line.startswith("s_endpgm"):
            hot = None
            continue
        w = WAIT.search(line)
        if w:
            n = int(w.group(1))
            hot = {r: v for r, v in hot.items() if v[1] < n}
            continue
        used = regs_of(line)
        clash = used & hot.keys()
        if clash:
            bad.append((no, raw.strip(), sorted(clash), sorted({hot[r][0] for r in clash})))
        if VMEM.match(line) is not None and "_load" in line.split()[0]:
            hot = {r: (l, y + 1) for r, (l, y) in hot.items()}
            if in_asm:
                first = line.split(None, 1)[1].split(",")[0]
                for r in regs_of(first):
                    hot[r] = (no, 0)
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
