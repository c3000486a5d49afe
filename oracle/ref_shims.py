"""TEST INFRASTRUCTURE ONLY -- never imported by the product path (tsdiff_amd/).

Import shims that let the *unchanged* reference sources under /root/reference
import and run on torch-CPU in the build container, where torch_geometric,
torch_scatter, torch_sparse, torch_cluster, rdkit, easydict, torchvision and
wandb are absent.

Only `oracle/gen_golden.py` uses this module, and only in the build container: /root/reference does not exist on the
GPU box.  The arithmetic that the reference delegates to third-party wheels is
restated here from their published semantics (pinned versions, reference
env.yaml:191-195):

  torch_geometric 1.7.2  to_dense_adj / dense_to_sparse / MessagePassing / radius_graph
  torch_scatter   2.0.8  scatter / scatter_add / scatter_mean
  torch_sparse    0.6.12 coalesce
  torch_cluster   1.5.9  radius_graph (strict `dist^2 < r^2`, max_num_neighbors=32)
  rdkit 2020.09.1        len(BondType.names) == 22   (reference utils/chem.py:21)

Everything else that the reference imports at module scope but never touches
on the hot path is a placeholder object.
"""
import importlib.abc
import importlib.machinery
import inspect
import sys
import types

import torch

REFERENCE_ROOT = "/root/reference"

_PLACEHOLDER_ROOTS = (
    "rdkit", "torchvision", "torch_geometric", "torch_scatter", "torch_sparse",
    "torch_cluster", "easydict", "wandb", "ase", "py3Dmol",
)


class _Placeholder:
    """Attribute sink: any attribute / call yields another placeholder."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Placeholder()

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _Placeholder()

    def __mro_entries__(self, bases):  # allow `class X(placeholder)`
        return (object,)


class _PlaceholderModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        cls = type(name, (_PlaceholderClass,), {})
        setattr(self, name, cls)
        return cls


class _PlaceholderClass:
    """A real class (so it can be subclassed / used in annotations)."""

    def __init__(self, *a, **k):
        pass


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in _PLACEHOLDER_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _PlaceholderModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


# --------------------------------------------------------------------------
# functional stand-ins
# --------------------------------------------------------------------------
class EasyDict(dict):
    """easydict 1.9: attribute access, recursive conversion of nested dicts."""

    def __init__(self, d=None, **kwargs):
        super().__init__()
        d = dict(d or {})
        d.update(kwargs)
        for k, v in d.items():
            setattr(self, k, v)

    def __setattr__(self, name, value):
        if isinstance(value, (list, tuple)):
            value = type(value)(EasyDict(x) if isinstance(x, dict) else x for x in value)
        elif isinstance(value, dict) and not isinstance(value, EasyDict):
            value = EasyDict(value)
        super().__setattr__(name, value)
        super().__setitem__(name, value)

    __setitem__ = __setattr__


EasyDict.__module__ = "easydict"  # pickles (checkpoints) must name the class as the real package does


def scatter(src, index, dim=0, out=None, dim_size=None, reduce="sum"):
    assert dim == 0
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() else 0
    shape = (dim_size,) + tuple(src.shape[1:])
    res = torch.zeros(shape, dtype=src.dtype, device=src.device)
    res.index_add_(0, index, src)
    if reduce in ("sum", "add"):
        return res
    if reduce == "mean":
        cnt = torch.zeros(dim_size, dtype=src.dtype, device=src.device)
        cnt.index_add_(0, index, torch.ones_like(index, dtype=src.dtype))
        cnt = cnt.clamp(min=1)
        return res / cnt.view((-1,) + (1,) * (src.dim() - 1))
    raise NotImplementedError(reduce)


def scatter_add(src, index, dim=0, out=None, dim_size=None):
    return scatter(src, index, dim=dim, dim_size=dim_size, reduce="sum")


def scatter_mean(src, index, dim=0, out=None, dim_size=None):
    return scatter(src, index, dim=dim, dim_size=dim_size, reduce="mean")


def to_dense_adj(edge_index, batch=None, edge_attr=None, max_num_nodes=None):
    """PyG 1.7.2 to_dense_adj for batch=None: (1, N, N[, F]); duplicates summed;
    edge_attr=None -> float32 ones."""
    assert batch is None
    N = max_num_nodes
    if N is None:
        N = int(edge_index.max()) + 1 if edge_index.numel() else 0
    if edge_attr is None:
        edge_attr = torch.ones(edge_index.size(1), device=edge_index.device)
    size = [1, N, N] + list(edge_attr.shape[1:])
    adj = torch.zeros(size, dtype=edge_attr.dtype, device=edge_index.device)
    flat = adj.view([N * N] + list(edge_attr.shape[1:]))
    flat.index_add_(0, edge_index[0] * N + edge_index[1], edge_attr)
    return adj


def dense_to_sparse(adj):
    """PyG 1.7.2 dense_to_sparse: nonzero in row-major order."""
    assert adj.dim() in (2, 3)
    index = adj.nonzero(as_tuple=True)
    edge_attr = adj[index]
    if len(index) == 3:
        batch = index[0] * adj.size(-1)
        index = (batch + index[1], batch + index[2])
    return torch.stack(index, dim=0), edge_attr


def coalesce(index, value, m, n, op="add"):
    """torch_sparse 0.6.12 coalesce: sort by (row, col), reduce duplicates."""
    key = index[0] * n + index[1]
    uniq, inv = torch.unique(key, sorted=True, return_inverse=True)
    new_index = torch.stack([uniq // n, uniq % n], dim=0)
    if value is None:
        return new_index, None
    new_val = torch.zeros((uniq.numel(),) + tuple(value.shape[1:]), dtype=value.dtype)
    new_val.index_add_(0, inv, value)
    return new_index, new_val


def radius_graph(x, r, batch=None, loop=False, max_num_neighbors=32,
                 flow="source_to_target", num_workers=1):
    """torch_cluster 1.5.9 radius_graph: same-graph pairs with dist^2 < r^2.

    The library truncates each query's neighbour list at `max_num_neighbors`;
    which neighbours survive is implementation defined, so the shim asserts the
    cap never binds instead of guessing (it cannot bind for <= 33-atom graphs).
    Output: edge_index[0] = source (neighbour), edge_index[1] = target (query),
    grouped by target.  The reference only ever uses it as an unordered set
    (sparse add + coalesce, reference models/common.py:341-382).
    """
    assert flow == "source_to_target"
    N = x.size(0)
    if batch is None:
        batch = torch.zeros(N, dtype=torch.long)
    diff = x.unsqueeze(0) - x.unsqueeze(1)
    d2 = (diff * diff).sum(-1)
    ok = (d2 < r * r) & (batch.unsqueeze(0) == batch.unsqueeze(1))
    if not loop:
        ok &= ~torch.eye(N, dtype=torch.bool)
    assert int(ok.sum(1).max()) <= max_num_neighbors if N else True, \
        "radius_graph max_num_neighbors cap would bind; shim undefined there"
    tgt, src = ok.nonzero(as_tuple=True)
    return torch.stack([src, tgt], dim=0)


class MessagePassing(torch.nn.Module):
    """PyG 1.7.2 MessagePassing restricted to what the reference uses:
    aggr='add', flow='source_to_target', node_dim=0 (-2 for 2-D features)."""

    def __init__(self, aggr="add", flow="source_to_target", node_dim=-2):
        super().__init__()
        assert aggr == "add" and flow == "source_to_target"
        self.aggr = aggr
        self.flow = flow
        self.node_dim = node_dim

    def propagate(self, edge_index, size=None, **kwargs):
        params = list(inspect.signature(self.message).parameters)
        src, dst = edge_index[0], edge_index[1]
        n_dst = None
        args = {}
        for name in params:
            if name.endswith("_j") or name.endswith("_i"):
                data = kwargs[name[:-2]]
                idx = src if name.endswith("_j") else dst
                if isinstance(data, (tuple, list)):
                    data = data[0] if name.endswith("_j") else data[1]
                if name.endswith("_i") or n_dst is None:
                    n_dst = data.size(0)
                args[name] = data.index_select(0, idx)
            else:
                args[name] = kwargs[name]
        for v in kwargs.values():
            if isinstance(v, (tuple, list)) and v[1] is not None:
                n_dst = v[1].size(0)
        if size is not None and size[1] is not None:
            n_dst = size[1]
        msg = self.message(**args)
        out = torch.zeros((n_dst,) + tuple(msg.shape[1:]), dtype=msg.dtype)
        out.index_add_(0, dst, msg)
        return self.update(out)

    def message(self, x_j):
        return x_j

    def update(self, inputs):
        return inputs


class GaussianSmearing(torch.nn.Module):
    """PyG schnet.GaussianSmearing (only imported, never on the hot path)."""

    def __init__(self, start=0.0, stop=5.0, num_gaussians=50):
        super().__init__()
        offset = torch.linspace(start, stop, num_gaussians)
        self.coeff = -0.5 / (offset[1] - offset[0]).item() ** 2
        self.register_buffer("offset", offset)

    def forward(self, dist):
        dist = dist.view(-1, 1) - self.offset.view(1, -1)
        return torch.exp(self.coeff * torch.pow(dist, 2))


class _BondTypeNames(dict):
    pass


_installed = False


def install():
    """Register the stand-ins, then put the reference on sys.path."""
    global _installed
    if _installed:
        return
    finder = _Finder()
    sys.meta_path.insert(0, finder)

    import importlib

    def mod(name):
        return importlib.import_module(name)

    mod("easydict").EasyDict = EasyDict
    ts = mod("torch_scatter")
    ts.scatter, ts.scatter_add, ts.scatter_mean = scatter, scatter_add, scatter_mean
    mod("torch_sparse").coalesce = coalesce
    mod("torch_cluster").radius_graph = radius_graph
    tgu = mod("torch_geometric.utils")
    tgu.to_dense_adj, tgu.dense_to_sparse = to_dense_adj, dense_to_sparse
    tgn = mod("torch_geometric.nn")
    tgn.radius_graph = radius_graph
    tgn.MessagePassing = MessagePassing
    mod("torch_geometric.nn.conv").MessagePassing = MessagePassing
    mod("torch_geometric.nn.models.schnet").GaussianSmearing = GaussianSmearing
    typing_mod = mod("torch_geometric.typing")
    for n in ("Adj", "OptTensor", "OptPairTensor", "Size", "Tensor"):
        setattr(typing_mod, n, object)
    # rdkit: BondType.names must have 22 entries (reference utils/chem.py:21)
    rdchem = mod("rdkit.Chem.rdchem")
    bt = type("BondType", (), {})
    bt.names = _BondTypeNames((f"BT{i}", f"BT{i}") for i in range(22))
    rdchem.BondType = bt
    rdl = mod("rdkit.RDLogger")
    rdl.DisableLog = lambda *a, **k: None
    mod("rdkit").RDLogger = rdl

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    _installed = True


def import_reference():
    """Returns the reference's own modules, imported unchanged."""
    install()
    import importlib
    epsnet = importlib.import_module("models.epsnet")
    sampler = importlib.import_module("models.sampler")
    geometry = importlib.import_module("models.geometry")
    common = importlib.import_module("models.common")
    return types.SimpleNamespace(
        get_model=epsnet.get_model, sampler=sampler, geometry=geometry, common=common,
        EasyDict=EasyDict,
    )
