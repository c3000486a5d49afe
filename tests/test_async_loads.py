"""tools/check_async_loads.py: the static check that no instruction of a compiled kernel touches the destination
registers of an inline-asm global load before the wait that covers it (csrc/split16.hpp, csrc/common.hpp: the
compiler believes such a register written at the asm statement and may copy or spill it under register pressure)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("check_async_loads", os.path.join(ROOT, "tools", "check_async_loads.py"))
chk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(chk)

HEAD = "_Z4kernPf:\n"
TAIL = "\ts_endpgm\n\t.section\t.rodata\n"


def run(body, tmp_path):
    p = tmp_path / "k.s"
    p.write_text(HEAD + body + TAIL)
    items = []
    return chk.main([str(p)])


def test_clean_ring_passes(tmp_path):
    body = """
\t;;#ASMSTART
\tglobal_load_dwordx4 v[10:13], v1, s[2:3]
\t;;#ASMEND
\t;;#ASMSTART
\tglobal_load_dwordx4 v[14:17], v1, s[4:5]
\t;;#ASMEND
\tv_add_f32_e32 v2, v3, v4
\t;;#ASMSTART
\ts_waitcnt vmcnt(1)
\t;;#ASMEND
\tv_mfma_f32_32x32x16_f16 v[20:35], v[6:9], v[10:13], v[20:35]
\ts_waitcnt vmcnt(0)
\tv_mfma_f32_32x32x16_f16 v[20:35], v[6:9], v[14:17], v[20:35]
"""
    assert run(body, tmp_path) == 0


def test_copy_before_the_wait_is_reported(tmp_path):
    body = """
\t;;#ASMSTART
\tglobal_load_dwordx4 v[10:13], v1, s[2:3]
\t;;#ASMEND
\tv_mov_b32_e32 v40, v11
\t;;#ASMSTART
\ts_waitcnt vmcnt(0)
\t;;#ASMEND
"""
    assert run(body, tmp_path) == 1


def test_spill_before_the_wait_is_reported(tmp_path):
    body = """
\t;;#ASMSTART
\tglobal_load_dwordx4 v[10:13], v1, s[2:3]
\t;;#ASMEND
\tscratch_store_dwordx4 off, v[10:13], off
\ts_waitcnt vmcnt(0)
"""
    assert run(body, tmp_path) == 1


def test_counted_wait_covers_only_the_older_loads(tmp_path):
    body = """
\t;;#ASMSTART
\tglobal_load_dwordx4 v[10:13], v1, s[2:3]
\t;;#ASMEND
\t;;#ASMSTART
\tglobal_load_dwordx4 v[14:17], v1, s[4:5]
\t;;#ASMEND
\t;;#ASMSTART
\ts_waitcnt vmcnt(1)
\t;;#ASMEND
\tv_mov_b32_e32 v40, v14
"""
    assert run(body, tmp_path) == 1   # the second load is still in flight behind vmcnt(1)


def test_a_pending_store_makes_the_counted_wait_weaker(tmp_path):
    # loads return in order among loads and stores among stores, not with respect to each other: with a store in
    # flight vmcnt(1) proves nothing about the one load
    body = """
\tglobal_store_dword v1, v2, s[0:1]
\t;;#ASMSTART
\tglobal_load_dwordx4 v[10:13], v1, s[2:3]
\t;;#ASMEND
\ts_waitcnt vmcnt(1)
\tv_mov_b32_e32 v40, v10
"""
    assert run(body, tmp_path) == 1


def test_compiler_issued_loads_are_the_compilers_business(tmp_path):
    body = """
\tglobal_load_dwordx4 v[10:13], v1, s[2:3]
\tv_mov_b32_e32 v40, v50
\ts_waitcnt vmcnt(0)
\tv_mov_b32_e32 v41, v10
"""
    assert run(body, tmp_path) == 0


def test_the_built_kernels_pass():
    """`make` (build()) runs the check over every translation unit with inline asm and keeps the listings"""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "tsdiff_amd", "csrc", "*.verify.s")))
    if not files:
        import pytest
        pytest.skip("no kernel listings (run `make -C tsdiff_amd/csrc`)")
    assert chk.main(files) == 0


def test_a_block_reached_only_from_before_the_loads_is_clean(tmp_path):
    # if (c) { loads } else { the compiler zero-fills the same registers }: the else block never sees the loads
    body = """
\ts_cbranch_scc1 .LBB0_2
\t;;#ASMSTART
\tglobal_load_dwordx4 v[10:13], v1, s[2:3]
\t;;#ASMEND
\ts_branch .LBB0_3
.LBB0_2:
\tv_mov_b32_e32 v10, 0
.LBB0_3:
\t;;#ASMSTART
\ts_waitcnt vmcnt(0)
\t;;#ASMEND
\tv_mov_b32_e32 v40, v10
"""
    assert run(body, tmp_path) == 0


def test_a_path_that_skips_the_wait_is_reported(tmp_path):
    body = """
\t;;#ASMSTART
\tglobal_load_dwordx4 v[10:13], v1, s[2:3]
\t;;#ASMEND
\ts_cbranch_scc1 .LBB0_2
\t;;#ASMSTART
\ts_waitcnt vmcnt(0)
\t;;#ASMEND
.LBB0_2:
\tv_mov_b32_e32 v40, v10
"""
    assert run(body, tmp_path) == 1
