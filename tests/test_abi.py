"""CPU: the C-ABI library loads and exports every symbol include/tsdiff_hip.h declares; host-side
logic that needs no GPU (config mapping, state_dict key compatibility, error mapping)."""
import ctypes as C
import os
import re

import pytest
import torch

from tests.util import ROOT


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "tsdiff_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(tsd_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from tsdiff_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = _lib.load()
    names = _declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/tsdiff_hip.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature in tsdiff_amd/_lib.py"
    assert set(_lib.SIGNATURES) == set(names)
    assert b"gfx950" in lib.tsd_version()


def test_weight_sizes_match_reference_parameter_count():
    from tsdiff_amd import _lib, engine, synth
    lib = _lib.load()
    cfg = engine.make_cfg(synth.DEFAULT_MODEL_CONFIG)
    # 2 770 305 trainable fp32 parameters in the reference model
    assert lib.tsd_raw_weight_floats(C.byref(cfg)) == 2770305
    assert lib.tsd_packed_weight_floats(C.byref(cfg)) >= 2770305
    bad = engine.make_cfg(synth.DEFAULT_MODEL_CONFIG)
    bad.hidden = 100
    assert lib.tsd_raw_weight_floats(C.byref(bad)) == 0
    assert b"unsupported" in lib.tsd_last_error()


def test_state_dict_keys_match_reference():
    """keys / shapes recorded from the reference model, including aliases"""
    from tsdiff_amd import synth
    from tsdiff_amd.epsnet import get_model
    from tsdiff_amd.utils import AttrDict
    model = get_model(AttrDict(synth.DEFAULT_MODEL_CONFIG))
    sd = model.state_dict()
    shapes = synth.param_shapes(synth.DEFAULT_MODEL_CONFIG)
    for k, (shape, _) in shapes.items():
        assert tuple(sd[k].shape) == tuple(shape), k
    assert tuple(sd["betas"].shape) == (5000,) and tuple(sd["alphas"].shape) == (5000,)
    for alias in ("model_embedding.0.weight", "model_embedding.1.weight", "model.0.bond_emb.weight",
                  "model.0.mlp.layers.1.bias", "model.1.interactions.6.conv.nn.2.weight",
                  "model.2.layers.2.weight"):
        assert alias in sd, alias
    n_train = sum(p.numel() for p in model.parameters() if p.requires_grad)
    assert n_train == 2770305
    with pytest.raises(NotImplementedError):
        get_model(AttrDict({**synth.DEFAULT_MODEL_CONFIG, "network": "nope"}))


def test_product_path_has_no_cpu_fallback():
    from tsdiff_amd import _lib, synth
    from tsdiff_amd.epsnet import get_model
    from tsdiff_amd.utils import AttrDict
    model = get_model(AttrDict(synth.small_model_config(64, 2)))
    b = synth.wb97xd3_like_batch(2, seed=0)
    t = {k: torch.from_numpy(v) for k, v in b.items() if hasattr(v, "shape")}
    with pytest.raises(_lib.TsdError):
        model(t["atom_type"], t["r_feat"], t["p_feat"], t["pos"], t["bond_index"], t["bond_type"], t["batch"],
              torch.zeros(2, dtype=torch.long))


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "tsdiff_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in src.replace("# checker", ""), f"{f} mentions the oracle"


def test_reference_checkpoint_round_trip(tmp_path):
    """a checkpoint written by the reference's torch.save (fixture from oracle/gen_golden.py) loads into the
    drop-in model: config via AttrDict, every state_dict key including the aliases (strict=True)"""
    import numpy as np
    from tsdiff_amd import io, synth
    from tsdiff_amd.epsnet import get_model
    ck = io.load_checkpoint(os.path.join(ROOT, "tests", "golden", "ckpt_small.pt"))
    assert ck["iteration"] == 1000 and ck["config"].model.network == "condensenc"
    model = get_model(ck["config"].model)
    missing, unexpected = model.load_state_dict(ck["model"], strict=True)
    assert not missing and not unexpected
    sd = synth.synth_state_dict(synth.small_model_config(64, 2), 1)
    for k, v in sd.items():
        assert np.array_equal(model.state_dict()[k].numpy(), v), k
    # and our own save format loads back the same way
    p = str(tmp_path / "ck.pt")
    opt = torch.optim.Adam(model.parameters(), lr=5e-4, betas=(0.95, 0.999))
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, factor=0.8, patience=10)
    io.save_checkpoint(p, ck["config"], model, opt, sched, iteration=7, avg_val_loss=0.5)
    ck2 = io.load_checkpoint(p)
    # attribute access like the reference's sampling.py:128 / train.py:113-118 (config is an EasyDict there)
    assert ck2["iteration"] == 7 and ck2["config"].model.hidden_dim == 64 and ck2["config"].train.batch_size == 200
    assert all(torch.equal(ck2["model"][k], model.state_dict()[k]) for k in model.state_dict())
    opt.load_state_dict(ck2["optimizer"])
    sched.load_state_dict(ck2["scheduler"])
    with pytest.raises(ValueError):
        io.save_checkpoint(p, ck["config"], model, None, None)
    # on disk the config is the reference's class (a reader with easydict installed gets an EasyDict back) and
    # nothing in the file needs this package to be importable
    import zipfile
    with zipfile.ZipFile(p) as z:
        pk = z.read([n for n in z.namelist() if n.endswith("data.pkl")][0])
    assert b"easydict\nEasyDict" in pk and b"tsdiff_amd" not in pk


def test_header_is_plain_c(tmp_path):
    """include/tsdiff_hip.h must be consumable from C (cgo / JNI / ctypes generators): compile a C99
    translation unit that includes it and references every declared function"""
    import subprocess
    names = _declared_symbols()
    src = tmp_path / "abi.c"
    src.write_text('#include "tsdiff_hip.h"\n' + "void* table[] = {" + ", ".join(f"(void*){n}" for n in names) + "};\n")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src),
                    "-o", str(tmp_path / "abi.o")], check=True)


def test_forward_work_model_known_answers():
    """tsd_forward_work (host only): the SURVEY 8(d) flop model at H = 256, L = 7 against its closed form, and the
    aggregate's algorithmic bytes 1028 E + 2048 N + 4"""
    import ctypes as C
    from tsdiff_amd import _lib, engine, synth
    lib = _lib.load()
    cfg = engine.make_cfg(synth.DEFAULT_MODEL_CONFIG)
    w = _lib.Work()
    N, E_enc, E_out, E_diff, L = 1600, 26074, 24000, 1500, 7
    _lib.check(lib.tsd_forward_work(C.byref(cfg), N, E_enc, E_out, E_diff, C.byref(w)))
    # inference forward: Linear(1,H) + ONE folded H x H GEMM per embedded edge (type-sorted tiles); the training
    # forward keeps the reference's order: + edge_cat's 6 H^2 per embedded edge
    F = ((E_enc // 2) * (131584 + L * 262144) + E_enc * L * 512 + E_diff * 131584
         + (E_out // 2) * (327936 + 256) + N * (L * 393216 + 13000))
    assert w.flops_train_forward == F + ((E_enc // 2) + E_diff) * 393216
    F_ref = (E_enc * (131584 + 393216 + L * 262144 + L * 512) + E_out * (131584 + 393216 + 327936 + 256)
             + N * (L * 393216 + 13000))
    assert w.flops_executed == F and w.flops_reference == F_ref
    assert w.flops_block_launch == (L * ((E_enc // 2) * (4.0 * 65536 + 256) + E_enc * 512.0 + N * 6.0 * 65536)) / (L + 1)
    assert w.bytes_aggregate == 1028.0 * E_enc + 2048.0 * N + 4
    assert w.flops_edge_embed + w.flops_blocks + w.flops_pair_output + w.flops_other == w.flops_executed
    assert lib.tsd_forward_work(C.byref(cfg), -1, 0, 0, 0, C.byref(w)) != 0


def test_size_queries_for_every_supported_hidden_width():
    """the host-only size queries (no GPU call): workspace / parameter-vector sizes for hidden 64, 128 and 256 --
    widths whose halves are not MFMA shapes size their split scratch by other rules (a division by zero sat here)"""
    import ctypes as C
    from tsdiff_amd import _lib, engine, synth
    lib = _lib.load()
    prev = None
    for hidden in (64, 128, 256):
        cfg = engine.make_cfg(synth.small_model_config(hidden, 3))
        raw = lib.tsd_train_raw_floats(C.byref(cfg))
        assert raw > 3 * 3 * hidden * hidden and (prev is None or raw > prev)
        prev = raw
        sizes = [lib.tsd_train_workspace_floats(C.byref(cfg), N, P) for N, P in ((0, 0), (10, 90), (300, 6000), (3000, 90000))]
        assert sizes == sorted(sizes) and sizes[1] > 0
        assert lib.tsd_forward_workspace_floats(C.byref(cfg), 3000, 90000, 1) > 0
        assert lib.tsd_forward_workspace_floats(C.byref(cfg), 3000, 90000, 8) > lib.tsd_forward_workspace_floats(C.byref(cfg), 3000, 90000, 1)


def test_entry_points_run_inside_roctx_ranges(tmp_path):
    """the library resolves roctxRangePushA / roctxRangePop from the process image (what `rocprofv3 --marker-trace`
    preloads) and brackets its entry points with "tsd:<name>" ranges; without those symbols the ranges are no-ops
    (every other test).  A stand-in roctx library, preloaded into a child process, records what it is called with."""
    import subprocess
    import sys
    src = tmp_path / "fake_roctx.c"
    src.write_text(r'''
#include <stdio.h>
#include <stdlib.h>
static int depth = 0;
static void note(const char* s) { FILE* f = fopen(getenv("FAKE_ROCTX_LOG"), "a"); if (f) { fprintf(f, "%s\n", s); fclose(f); } }
int roctxRangePushA(const char* m) { char b[256]; snprintf(b, sizeof b, "push %d %s", depth, m); note(b); return depth++; }
int roctxRangePop(void) { char b[64]; snprintf(b, sizeof b, "pop %d", --depth); note(b); return depth; }
''')
    so = tmp_path / "libfake_roctx.so"
    subprocess.run(["gcc", "-shared", "-fPIC", "-O1", str(src), "-o", str(so)], check=True)
    log = tmp_path / "ranges.log"
    code = ("import ctypes as C, sys; sys.path.insert(0, %r); from tsdiff_amd import _lib, engine, synth; "
            "lib = _lib.load(); cfg = engine.make_cfg(synth.DEFAULT_MODEL_CONFIG); "
            "rc = lib.tsd_pack_weights(C.byref(cfg), None, None, None); print('rc', rc)" % str(ROOT))
    env = dict(os.environ, LD_PRELOAD=str(so), FAKE_ROCTX_LOG=str(log))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "rc -" in out.stdout, out.stdout + out.stderr   # null pointers: an error code, no crash
    lines = log.read_text().split("\n")
    assert "push 0 tsd:pack_weights" in lines and "pop 0" in lines
    assert sum(l.startswith("push") for l in lines) == sum(l.startswith("pop") for l in lines)


def test_options_object(monkeypatch):
    """tsdiff_amd.options: ONE object, defaults = the fast paths, TSDIFF_* environment variables read once at import
    (Options.from_env), bad values refused"""
    from tsdiff_amd import options
    o = options.Options()
    assert (o.gemm, o.train, o.train_gemm, o.train_side_lane, o.one_launch, o.fused_encoder) == ("h2", "fused", "h2", True, True, True)
    monkeypatch.setenv("TSDIFF_TRAIN_GEMM", "f32")
    monkeypatch.setenv("TSDIFF_TRAIN_SIDE_LANE", "0")
    monkeypatch.setenv("TSDIFF_TRAIN_FALLBACK_LATCH", "4")
    e = options.Options.from_env()
    assert e.train_gemm == "f32" and e.train_side_lane is False and e.train_fallback_latch == 4 and e.gemm == "h2"
    monkeypatch.setenv("TSDIFF_TRAIN_GEMM", "bf16")
    with pytest.raises(ValueError):
        options.Options.from_env()
    monkeypatch.setenv("TSDIFF_TRAIN_GEMM", "h2")
    monkeypatch.setenv("TSDIFF_TRAIN", "eager")
    with pytest.raises(ValueError):
        options.Options.from_env()
    # the documented table lists every field
    doc = options.__doc__
    for f in options.Options.__dataclass_fields__:
        assert f"| {f} " in doc, f
