#!/usr/bin/env python3
"""Register-budget guard of the compiled kernels (`make verify`).

Several kernels are built to a residency, not just to correctness: the one-launch forward and the block launches
must keep TWO workgroups of 8 waves per CU (<= 128 VGPRs, AGPRs included: gfx950 has one unified file), the typed embedding
tile THREE (<= 80), WITHOUT an occupancy cap, because under a cap the compiler spills beside the asm-issued load rings (docs/NOTEBOOK.md, "what went wrong on the way to the one-launch kernel"); the fused
per-unit encoder owns a CU (<= 256).  None of them may touch scratch memory.  A compiler bump that moves one of them
over its line halves the occupancy silently -- this check reads the `.amdhsa` metadata of the `.verify.s` files that
`make verify` already writes and fails the build instead.

usage: check_regs.py FILE.verify.s ...
"""
import re
import sys

# (H = 256 is the shipped configuration and the only one with 8-wave workgroups; H = 64 / 128 workgroups are 2 / 4 waves
# and LDS-limited, their slot count comes from the occupancy query of mega_slots())
# kernel-name regex (mangled names contain the template arguments) -> (max vgprs incl. agprs, scratch bytes allowed)
BUDGET = [
    (r"forward_mega_kernelILi256E", 128, 0),
    (r"typed_embed_h_kernelILi256E", 80, 0),   # THREE workgroups of 8 waves per CU (37 KB of LDS each)
    (r"layer_combo_kernelILi256ELb0ELb[01]ELi1E", 128, 0),  # split-f16 block launches (all filter-tile widths)
    (r"pair_output_h_kernelILi256ELb0ELi1E", 80, 0),  # inference form: THREE workgroups per CU
    (r"pair_output_h_kernelILi256ELb0ELi2E", 128, 0),  # ... its 64-row tiles (launches many rounds deep): two
    (r"pair_output_h_kernelILi256ELb1ELi1E", 128, 0),  # saving form of the training step
    (r"layer_combo_kernelILi256ELb1ELb0ELi1ELi1E", 128, 0),  # split-f16 block launch of the training step (saving form)
    (r"block_bwd_kernelILi256E", 128, 0),                    # backward block launch, fp32 and split-f16 filter chains
    (r"pair_bwd_h_kernelILi256E", 128, 0),                   # split-f16 training step: the other tile kernels
    (r"embed_bwd_h_kernelILi256E", 128, 0),
    (r"edge_embed_save_h_kernelILi256E", 128, 0),
    (r"unit_encoder_kernelILi256E", 256, 0),
    (r"wgrad_h2_batch_kernel", 256, 0),  # four waves per workgroup, two workgroups per CU
]


def kernels(path):
    """yield (name, dict) for every kernel metadata record of an assembly listing"""
    name, rec = None, {}
    with open(path) as f:
        for line in f:
            m = re.match(r"\s*-?\s*\.(\w+):\s*(\S+)\s*$", line)
            if not m:
                continue
            k, v = m.group(1), m.group(2)
            if k in ("agpr_count", "args") and "name" in rec and "vgpr_count" in rec:
                yield rec["name"], rec
                rec = {}
            if k == "name" and v.startswith("_Z"):
                rec["name"] = v
            elif k in ("vgpr_count", "agpr_count", "private_segment_fixed_size", "sgpr_spill_count", "vgpr_spill_count",
                       "sgpr_count", "group_segment_fixed_size"):
                rec[k] = int(v)
    if "name" in rec and "vgpr_count" in rec:
        yield rec["name"], rec


def main(paths):
    seen = {pat: 0 for pat, _, _ in BUDGET}
    bad = 0
    for path in paths:
        for name, rec in kernels(path):
            for pat, vmax, smax in BUDGET:
                if not re.search(pat, name):
                    continue
                seen[pat] += 1
                v = rec.get("vgpr_count", 0)
                scr = rec.get("private_segment_fixed_size", 0)
                spill = rec.get("vgpr_spill_count", 0)
                ok = v <= vmax and scr <= smax and (spill == 0 or smax > 0)
                print(f"check_regs: {'ok  ' if ok else 'FAIL'} {v:3d}/{vmax} VGPRs, scratch {scr} B, "
                      f"{rec.get('sgpr_spill_count', 0)} SGPR spills  {name[:90]}")
                if not ok:
                    bad += 1
    unit = any("kernels_unit" in p for p in paths)
    others = any("kernels_combo" in p for p in paths)
    missing = [pat for pat, n in seen.items() if n == 0 and (unit if pat.startswith("unit_encoder") else others)]
    for pat in missing:
        print(f"check_regs: FAIL no compiled kernel matches {pat!r} (renamed? update tools/check_regs.py)")
    return 1 if bad or missing else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
