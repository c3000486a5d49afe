"""Mirror of the graph-extension entry points of the reference `models/common.py:205-223` and
`models/epsnet/condensenc.py:117-154`, served by the device-side topology + geometry kernels."""
import torch

from . import engine


def extend_condensed_graph_edge(pos, bond_index, bond_type, batch, edge_order=4, cutoff=10.0,
                                atom_type=None):
    """-> (edge_index (2,E) int64 row-major sorted, edge_type_r (E,), edge_type_p (E,), edge_length (E,))
    == reference CondenseEncoderEpsNetwork._extend_condensed_graph_edge for `edge_order` plus
    get_distance on the result."""
    N = int(pos.shape[0])
    cfg = engine.ModelCfg(hidden=64, num_convs=1, feat_dim=1, edge_order=int(edge_order),
                          pred_edge_order=int(edge_order), edge_cutoff=float(cutoff), conv_cutoff=float(cutoff),
                          smooth_conv=0)
    at = atom_type if atom_type is not None else torch.zeros(N, dtype=torch.int64, device=pos.device)
    feat = torch.zeros(N, 1, dtype=torch.int64, device=pos.device)
    db = engine.DeviceBatch(cfg, at, feat, feat, bond_index, bond_type, batch)
    db.geometry(pos)
    ei, el, tr, tp = db.edges_to_torch("out")
    return ei, tr, tp, el.view(-1)
