"""CPU tests of bench.py's host-side contract pieces (no GPU): the key order of the `roofline` object, the options object."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def test_roofline_keys_scalars_first_prose_as_one_string_objects_last():
    """the driver's record keeps about two dozen keys of `roofline`: the contract's eight, then the flat scalars of
    the other sections, before anything droppable"""
    import bench
    r = {"kernel": "k", "bound": "mfma", "achieved": 1.0, "peak": 2.0, "unit": "TFLOP/s", "frac": 0.5,
         "peak_note": "why this peak", "traffic": None, "avg_launch_us": 3.0, "flop_per_launch": 9, "bound_note": "why this bound",
         "traffic_source": "file", "f32_mfma": {"frac": 0.6}, "aggregate": {"frac": 0.8}, "c5_frac": 0.3, "train_ms_per_step": 1.8,
         "aggregate_frac": 0.8, "workload": "w"}
    o = bench.order_roofline(r)
    keys = list(o)
    assert keys[:8] == ["kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us"]
    for name in ("f32_ms_per_step", "f32_frac", "ensemble8_ms_per_step", "c5_ms_per_step", "c5_frac", "train_ms_per_step",
                 "train_frac", "aggregate_frac", "cold_ms_per_step"):
        assert name in keys[:24], name
    assert o["c5_frac"] == 0.3 and o["train_ms_per_step"] == 1.8 and o["aggregate_frac"] == 0.8
    assert "peak_note" not in o and "bound_note" not in o and "why this peak" in o["notes"] and "why this bound" in o["notes"]
    first_obj = min(i for i, k in enumerate(keys) if isinstance(o[k], dict))
    assert all(not isinstance(o[k], dict) for k in keys[:first_obj]) and all(isinstance(o[k], dict) for k in keys[first_obj:])
    assert keys.index("notes") < first_obj


def test_options_read_the_environment_once_and_validate(monkeypatch):
    from tsdiff_amd import options
    monkeypatch.setenv("TSDIFF_DP_OVERLAP", "1")
    monkeypatch.setenv("TSDIFF_TRAIN_FLAT_GRAD", "0")
    o = options.Options.from_env()
    assert o.dp_overlap is True and o.train_flat_grad is False and o.gemm == "h2"
    monkeypatch.delenv("TSDIFF_DP_OVERLAP")
    monkeypatch.delenv("TSDIFF_TRAIN_FLAT_GRAD")
    o = options.Options.from_env()
    assert o.dp_overlap is False and o.train_flat_grad is True
    monkeypatch.setenv("TSDIFF_GEMM", "bf16")
    import pytest
    with pytest.raises(ValueError):
        options.Options.from_env()
