"""On-disk formats of the reference.

* checkpoints (`train.py:221-231`): `torch.save({"config": EasyDict, "model": state_dict, "optimizer": ...,
  "scheduler": ..., "iteration": int, "avg_val_loss": float})`.  `easydict` is not a dependency here: pickled
  `easydict.EasyDict` objects are read as `tsdiff_amd.utils.AttrDict`, and AttrDict is WRITTEN as
  `easydict.EasyDict`, so files from either side load on the other (`ckpt["config"].model` works on both).
* sampling results (`sampling.py:218-243`): a pickled list of torch_geometric (1.7.2) `Data` objects, one per
  sampled graph, each with the input fields (`atom_type, r_feat, p_feat, edge_index, edge_type, pos, smiles,
  rdmol, ...`) plus `pos_gen` -- (n, 3) final positions, or (n_steps, n, 3) with `--save_traj`, where the
  trajectory is scaled by sqrt(alpha) of its step (`sampling.py:210-216`).  PyG 1.7 pickles a `Data` as its
  attribute dict, so the list is read here into `SampleRecord` objects (same attributes) and written back under
  the class name `torch_geometric.data.data.Data`; objects of libraries that are absent here (rdkit `Mol`) pass
  through as opaque constructor-argument / state blobs and are written back unchanged.
"""
import io as _io
import os
import pickle
import types

import torch

from .utils import AttrDict

_DATA_CLASS = ("torch_geometric.data.data", "Data")
_FOREIGN_ROOTS = ("torch_geometric", "rdkit", "networkx")


class SampleRecord:
    """One sampled graph: a plain attribute bag with the attribute dict of a PyG 1.7 `Data` (sampling.py:218-226)."""
    _pickle_as = _DATA_CLASS

    def __init__(self, **kw):
        self.__dict__.update(kw)

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)

    def __getstate__(self):
        return self.__dict__

    def keys(self):
        return [k for k, v in self.__dict__.items() if v is not None]

    def to(self, device):
        for k, v in self.__dict__.items():
            if torch.is_tensor(v):
                self.__dict__[k] = v.to(device)
        return self

    def __repr__(self):
        items = ", ".join(f"{k}={list(v.shape) if torch.is_tensor(v) else type(v).__name__}"
                          for k, v in self.__dict__.items() if v is not None)
        return f"SampleRecord({items})"


class _Opaque:
    """Object of a class that cannot be imported here (e.g. rdkit.Chem.rdchem.Mol): keeps what its pickle held --
    constructor arguments and state -- and is written back under its original class name."""
    _pickle_as = None
    _args = ()
    _state = None

    def __init__(self, *args):
        self._args = args
        self._state = None

    def __setstate__(self, state):
        self._state = state

    def __reduce__(self):
        return (type(self), self._args) if self._state is None else (type(self), self._args, self._state)


_opaque_classes = {}


def _opaque_class(module, name):
    key = (module, name)
    if key not in _opaque_classes:
        _opaque_classes[key] = type(name, (_Opaque,), {"_pickle_as": key})
    return _opaque_classes[key]


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        root = module.split(".")[0]
        if root == "easydict":
            return AttrDict
        if (module, name) == _DATA_CLASS:
            return SampleRecord
        if root in _FOREIGN_ROOTS:
            # never imported, installed or not: reading a result file must not depend on rdkit / PyG versions.
            # (With rdkit available: Chem.Mol(rec.rdmol[0]._args[0]) rebuilds the molecule from its binary blob.)
            return _opaque_class(module, name)
        return super().find_class(module, name)


class _Pickler(pickle._Pickler):
    """Writes AttrDict / SampleRecord / opaque pass-through objects under the REFERENCE's class names (pure-Python
    pickler: the C one cannot rename a global)."""

    def save_global(self, obj, name=None):
        alias = ("easydict", "EasyDict") if obj is AttrDict else getattr(obj, "_pickle_as", None)
        if isinstance(obj, type) and alias:
            self.write(pickle.GLOBAL + alias[0].encode() + b"\n" + alias[1].encode() + b"\n")
            self.memoize(obj)
            return
        super().save_global(obj, name)


_pickle_module = types.SimpleNamespace(
    Unpickler=_Unpickler, load=lambda f, **kw: _Unpickler(f, **kw).load(), __name__="tsdiff_amd_pickle",
    loads=lambda b, **kw: _Unpickler(_io.BytesIO(b), **kw).load(),
    dump=lambda obj, f, protocol=None: _Pickler(f, protocol).dump(obj),
    dumps=pickle.dumps, Pickler=_Pickler)


# ---------------------------------------------------------------------------------------------------
# checkpoints
# ---------------------------------------------------------------------------------------------------
def load_checkpoint(path, map_location="cpu"):
    """Reference checkpoint -> dict with `config` as AttrDict (so `get_model(ckpt["config"].model)` and
    `model.load_state_dict(ckpt["model"])` work exactly as in `sampling.py:124-132`).  Files written by
    `save_checkpoint` load the same way; a plain-dict config (older files of this repo) is wrapped."""
    ckpt = torch.load(path, map_location=map_location, weights_only=False, pickle_module=_pickle_module)
    if isinstance(ckpt, dict) and isinstance(ckpt.get("config"), dict) and not isinstance(ckpt["config"], AttrDict):
        ckpt["config"] = AttrDict(ckpt["config"])
    return ckpt


def save_checkpoint(path, config, model, optimizer, scheduler, iteration=0, avg_val_loss=None):
    """Same dictionary as `train.py:221-231`, the config stored as the reference stores it (an
    `easydict.EasyDict` on disk; pass a dict / AttrDict).  `optimizer` and `scheduler` are required like there:
    the reference's resume path (`train.py:113-118`) loads both."""
    if optimizer is None or scheduler is None:
        raise ValueError("save_checkpoint needs the optimizer and the scheduler (the reference's --resume_iter path "
                         "loads both, train.py:116-117)")
    torch.save({"config": AttrDict(config), "model": model.state_dict(), "optimizer": optimizer.state_dict(),
                "scheduler": scheduler.state_dict(), "iteration": iteration, "avg_val_loss": avg_val_loss},
               path, pickle_module=_pickle_module, pickle_protocol=2)


# ---------------------------------------------------------------------------------------------------
# sampling results
# ---------------------------------------------------------------------------------------------------
def unbatch_positions(pos_gen, batch, num_graphs):
    """`sampling.py:218-223`: per-graph position tensors, original order, on `pos_gen`'s device like the
    reference's `pos_gen[batch == j]` mask loop (ids outside [0, num_graphs) are ignored like there).  `batch`
    ascending (what `Batch.from_data_list` produces): each graph is one contiguous slice, cut with one split
    instead of `num_graphs` mask kernels; anything else takes the literal mask loop."""
    pos_gen = pos_gen.detach()
    b = batch.detach().to(pos_gen.device)
    ok = b.dtype in (torch.int64, torch.int32) and b.numel() > 0
    if ok:
        bl = b.long()
        ok = bool(((bl[1:] >= bl[:-1]).all() & (bl[0] >= 0) & (bl[-1] < num_graphs)).item())
    if not ok:
        return [pos_gen[..., b == j, :] for j in range(num_graphs)]
    counts = torch.bincount(bl, minlength=num_graphs).tolist()
    return list(torch.split(pos_gen, counts, dim=-2))


def scale_trajectory(pos_gen_traj, alphas, n_steps, denoise_from_time_t=None):
    """`sampling.py:210-216` (`--save_traj`): the stacked trajectory, step k multiplied by sqrt(alpha) of the
    time index it was produced at (the loop runs the schedule slice backwards)."""
    alphas = alphas.detach().cpu()
    T = alphas.shape[0]
    hi = T if denoise_from_time_t is None else denoise_from_time_t
    a = alphas[hi - n_steps: hi].flip(0).view(-1, 1, 1)
    traj = pos_gen_traj if torch.is_tensor(pos_gen_traj) else torch.stack(list(pos_gen_traj))
    return traj.cpu() * a.sqrt()


def results_from_batch(data_list, batch, pos_gen, pos_gen_traj=None, save_traj=False, alphas=None,
                       denoise_from_time_t=None):
    """The per-graph records `sampling.py:218-226` appends for one sampled batch: `data_list` are the batch's
    input records (SampleRecord / any attribute object / dict, one per graph), copied with `pos_gen` added."""
    G = len(data_list)
    if save_traj:
        traj = scale_trajectory(pos_gen_traj, alphas, len(pos_gen_traj), denoise_from_time_t)
        parts = unbatch_positions(traj, batch, G)
    else:
        parts = unbatch_positions(pos_gen, batch, G)
    out = []
    for d, p in zip(data_list, parts):
        fields = dict(d) if isinstance(d, dict) else dict(d.__dict__)
        rec = SampleRecord(**fields)
        rec.pos_gen = p.clone()
        out.append(rec.to("cpu"))
    return out


def save_samples(path, results):
    """`pickle.dump(results, f)` of `sampling.py:229-231,241-243` with the records written as PyG `Data`."""
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        _Pickler(f, protocol=3).dump(list(results))
    os.replace(tmp, path)


def load_samples(path):
    """A `samples_all.pkl` / `samples_not_all.pkl` of the reference (or of `save_samples`) -> list of records."""
    with open(path, "rb") as f:
        return _Unpickler(f).load()


class ResultWriter:
    """The reference's result bookkeeping (`sampling.py:160-167,224-243`): results accumulate, the rolling
    `samples_not_all.pkl` is rewritten after every batch, `finish()` writes `samples_all.pkl` and removes the
    partial file; `resume=` continues from a partial file and `done_smiles` says which reactions it holds."""

    def __init__(self, log_dir, resume=None):
        self.log_dir = log_dir
        os.makedirs(log_dir, exist_ok=True)
        self.results = load_samples(resume) if resume is not None else []
        self.done_smiles = {getattr(r, "smiles", None) for r in self.results}
        self.partial_path = os.path.join(log_dir, "samples_not_all.pkl")

    def add_batch(self, records):
        for r in records:
            self.results.append(r)
            self.done_smiles.add(getattr(r, "smiles", None))
        save_samples(self.partial_path, self.results)

    def finish(self):
        if os.path.exists(self.partial_path):
            os.remove(self.partial_path)
        path = os.path.join(self.log_dir, "samples_all.pkl")
        save_samples(path, self.results)
        return path
