"""Drop-in for `GaussianSmearingEdgeEncoder` of the reference `models/encoder/edge.py:18-41`
(`edge_encoder: gaussian`; the literal radial-basis expansion)."""
import torch
from torch import nn

from .. import _lib
from .._lib import check, ptr, stream_ptr


class GaussianSmearingEdgeEncoder(nn.Module):
    def __init__(self, num_gaussians=64, cutoff=10.0):
        super().__init__()
        self.num_gaussians = num_gaussians
        self.cutoff = cutoff
        offset = torch.linspace(0.0, cutoff * 2, num_gaussians)  # schnet.py:14-19, stop = 2*cutoff (edge.py:26)
        self.coeff = -0.5 / (offset[1] - offset[0]).item() ** 2
        self.rbf = nn.Module()
        self.rbf.register_buffer("offset", offset)
        self.bond_emb = nn.Embedding(100, embedding_dim=num_gaussians)

    @property
    def out_channels(self):
        return self.num_gaussians * 2

    def forward(self, edge_length, edge_type):
        lib = _lib.load()
        if edge_length.device.type != "cuda":
            raise _lib.TsdError("tsdiff_amd runs on the GPU only (no CPU fallback)")
        d = edge_length.detach().to(torch.float32).contiguous().view(-1)
        t = edge_type.to(torch.int64).contiguous()
        K = self.num_gaussians
        out = torch.empty(d.shape[0], 2 * K, dtype=torch.float32, device=d.device)
        check(lib.tsd_gaussian_edge_encode(d.shape[0], K, float(self.coeff), ptr(d), ptr(self.rbf.offset),
                                           ptr(t), ptr(self.bond_emb.weight.detach()), ptr(out), stream_ptr()))
        return out
