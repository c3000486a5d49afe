"""Drop-in for the reference `models/encoder/gin.py` (GeoDiff legacy local encoder).

Same module / parameter names (`convs.{i}.nn.layers.{0,1}.{weight,bias}`, `convs.{i}.eps`,
`node_emb.weight`); the message passing is the HIP kernel `tsd_gine_aggregate`, the two dense layers of
every conv go through `tsd_linear_fwd`.  Inference only (the legacy network is not reachable from the
shipped train.py / sampling.py)."""
import ctypes as C

import torch
from torch import nn

from .. import _lib
from .._lib import check, ptr, stream_ptr

_ACT = {None: 0, "ReLU": 1, "Softplus": 2}
_ACT_KIND = {"ReLU": 2, "Softplus": 3}  # tsd_act_fwd kinds


class _MLP(nn.Module):
    def __init__(self, input_dim, hidden_dims):
        super().__init__()
        dims = [input_dim] + list(hidden_dims)
        self.layers = nn.ModuleList(nn.Linear(dims[i], dims[i + 1]) for i in range(len(dims) - 1))


def _linear(x, lin):
    lib = _lib.load()
    y = torch.empty(x.shape[0], lin.out_features, dtype=torch.float32, device=x.device)
    check(lib.tsd_linear_fwd(x.shape[0], lin.in_features, lin.out_features, ptr(x), ptr(lin.weight.detach()),
                             ptr(lin.bias.detach()), ptr(y), None, 0, stream_ptr()))
    return y


def _act(x, name):
    if name is None:
        return x
    lib = _lib.load()
    y = torch.empty_like(x)
    check(lib.tsd_act_fwd(_ACT_KIND[name], x.numel(), ptr(x), ptr(y), stream_ptr()))
    return y


class GINEConv(nn.Module):
    """reference gin.py:19-76"""

    def __init__(self, nn_module, eps=0.0, train_eps=False, activation="Softplus"):
        super().__init__()
        self.nn = nn_module
        self.initial_eps = eps
        self.activation = activation if isinstance(activation, str) else None
        if self.activation not in _ACT:
            raise NotImplementedError(f"GINEConv activation {activation}")
        if train_eps:
            self.eps = torch.nn.Parameter(torch.Tensor([eps]))
        else:
            self.register_buffer("eps", torch.Tensor([eps]))

    def forward(self, x, edge_index, edge_attr):
        lib = _lib.load()
        if x.device.type != "cuda":
            raise _lib.TsdError("tsdiff_amd runs on the GPU only (no CPU fallback)")
        x = x.detach().to(torch.float32).contiguous()
        ea = edge_attr.detach().to(torch.float32).contiguous()
        ei = edge_index.to(torch.int64).contiguous()
        assert x.size(-1) == ea.size(-1)
        out = torch.empty_like(x)
        check(lib.tsd_gine_aggregate(x.shape[0], ei.shape[1], x.shape[1], _ACT[self.activation], float(self.eps),
                                     ptr(x), ptr(ei), ptr(ea), ptr(out), stream_ptr()))
        h = _act(_linear(out, self.nn.layers[0]), self.nn.act_name)
        return _linear(h, self.nn.layers[1])


class GINEncoder(nn.Module):
    """reference gin.py:79-149"""

    def __init__(self, hidden_dim, num_convs=3, activation="ReLU", short_cut=True, concat_hidden=False,
                 embedding=False):
        super().__init__()
        self.hidden_dim, self.num_convs = hidden_dim, num_convs
        self.short_cut, self.concat_hidden, self.embedding = short_cut, concat_hidden, embedding
        if embedding:
            self.node_emb = nn.Embedding(100, hidden_dim)
        self.activation = activation if isinstance(activation, str) else None
        self.convs = nn.ModuleList()
        for _ in range(num_convs):
            mlp = _MLP(hidden_dim, [hidden_dim, hidden_dim])
            mlp.act_name = self.activation
            self.convs.append(GINEConv(mlp, activation=activation))

    def forward(self, z, edge_index, edge_attr):
        node_attr = self.node_emb.weight.detach()[z] if self.embedding else z  # gather: plumbing
        hiddens = []
        conv_input = node_attr.to(torch.float32).contiguous()
        for k, conv in enumerate(self.convs):
            hidden = conv(conv_input, edge_index, edge_attr)
            if k < len(self.convs) - 1 and self.activation is not None:
                hidden = _act(hidden, self.activation)
            if self.short_cut:
                hidden = hidden + conv_input
            hiddens.append(hidden)
            conv_input = hidden
        return torch.cat(hiddens, dim=-1) if self.concat_hidden else hiddens[-1]
