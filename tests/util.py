"""Shared helpers for the test-suite (golden loading, tolerance reports)."""
import json
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    d = {k: z[k] for k in z.files if k != "meta"}
    meta = json.loads(bytes(z["meta"]).decode())
    return d, meta


def batch_inputs(d, torch_=True):
    b = {k[3:]: d[k] for k in d if k.startswith("in_")}
    b["num_graphs"] = int(len(b["num_nodes_per_graph"]))
    if torch_:
        b = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in b.items()}
    return b


def rel_err(a, ref):
    """max |a-ref| / max|ref|  (per-tensor scale; the tolerance form of SURVEY.md 7.2)."""
    a = np.asarray(a, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    scale = max(np.abs(ref).max(), 1e-30) if ref.size else 1.0
    return float(np.abs(a - ref).max() / scale) if ref.size else 0.0


def rel_err_elementwise(a, ref, floor_frac=0.01):
    """max |a-ref| / |ref| over the entries with |ref| >= floor_frac * max|ref| (the per-element relative error where it
    means something: below the floor an entry's relative error is bounded through the per-tensor form instead)"""
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    ref = np.asarray(ref, dtype=np.float64).reshape(-1)
    if not ref.size:
        return 0.0
    big = np.abs(ref) >= floor_frac * np.abs(ref).max()
    return float((np.abs(a - ref)[big] / np.abs(ref)[big]).max()) if big.any() else 0.0


def assert_close(a, ref, rtol, what="", elem_tol=None):
    """per-tensor form max|d| / max|ref| <= rtol; `elem_tol`: additionally the per-element relative error
    of every entry above 1 % of the tensor's scale (parity caveat (a))"""
    a = np.asarray(a)
    ref = np.asarray(ref)
    assert a.shape == ref.shape, f"{what}: shape {a.shape} vs {ref.shape}"
    e = rel_err(a, ref)
    assert e <= rtol, f"{what}: max|d|/max|ref| = {e:.3e} > {rtol:.1e}"
    if elem_tol is not None:
        ee = rel_err_elementwise(a, ref)
        assert ee <= elem_tol, f"{what}: per-element |d|/|ref| = {ee:.3e} > {elem_tol:.1e} on entries above 1 % of the scale"
    return e
