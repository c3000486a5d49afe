"""On-disk formats of the reference.

* checkpoints (`train.py:221-231`): `torch.save({"config": EasyDict, "model": state_dict, "optimizer": ...,
  "scheduler": ..., "iteration": int, "avg_val_loss": float})`.  `easydict` is not a dependency here, so the
  pickled `easydict.EasyDict` objects are mapped onto `tsdiff_amd.utils.AttrDict` while unpickling.
* sampling results (`sampling.py:229-243`): a pickled list of PyG `Data`; `unbatch_positions` reproduces the
  `sampling.py:218-223` mask loop that produces each `data.pos_gen`.
"""
import pickle
import types

import torch

from .utils import AttrDict


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.split(".")[0] == "easydict":
            return AttrDict
        return super().find_class(module, name)


_pickle_module = types.SimpleNamespace(
    Unpickler=_Unpickler, load=lambda f, **kw: _Unpickler(f, **kw).load(), __name__="tsdiff_amd_pickle",
    loads=pickle.loads, dump=pickle.dump, dumps=pickle.dumps, Pickler=pickle.Pickler)


def load_checkpoint(path, map_location="cpu"):
    """Reference checkpoint -> dict with `config` as AttrDict (so `get_model(ckpt["config"].model)` and
    `model.load_state_dict(ckpt["model"])` work exactly as in `sampling.py:124-132`)."""
    return torch.load(path, map_location=map_location, weights_only=False, pickle_module=_pickle_module)


def save_checkpoint(path, config, model, optimizer=None, scheduler=None, iteration=0, avg_val_loss=None):
    """Same dictionary layout as `train.py:221-231` (config stored as a plain nested dict)."""
    def plain(x):
        if isinstance(x, dict):
            return {k: plain(v) for k, v in x.items()}
        if isinstance(x, (list, tuple)):
            return type(x)(plain(v) for v in x)
        return x
    torch.save({"config": plain(config), "model": model.state_dict(),
                "optimizer": optimizer.state_dict() if optimizer is not None else None,
                "scheduler": scheduler.state_dict() if scheduler is not None else None,
                "iteration": iteration, "avg_val_loss": avg_val_loss}, path)


def unbatch_positions(pos_gen, batch, num_graphs):
    """`sampling.py:218-223`: per-graph position tensors, original order."""
    return [pos_gen[batch == j] for j in range(num_graphs)]
