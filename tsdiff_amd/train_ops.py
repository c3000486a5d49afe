"""Differentiable building blocks of the training step (reference train.py:124-152,
models/epsnet/condensenc.py:267-328) as torch.autograd.Function nodes over the C-ABI training
primitives of include/tsdiff_hip.h.

torch's autograd engine only SEQUENCES the backward (so the reference's unmodified
`loss.mean().backward()`, `clip_grad_norm_` and `torch.optim.Adam` keep working); every forward and
backward computation below runs in libtsdiff_hip.so: the dense layers as fp32-MFMA tile kernels, the
graph-shaped operations (segmented aggregation and both adjoints, pair products, embedding
gather/scatter, the distance -> Cartesian chain rule, activations) as HIP kernels.

`get_loss` in training mode does NOT come through these nodes any more: FusedTrainLoss at the end of this file
hands the whole forward + loss and the whole backward to csrc/train_step.hip (one node).  The per-operation
nodes serve the differentiable `forward()`, the legacy dual-encoder network and the cross-check of the fused
step (`TSDIFF_TRAIN=ops`).
"""
import ctypes as C

import torch

from . import _lib
from .options import OPTIONS
from ._lib import check, ptr, stream_ptr


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


_SCRATCH = {}


def _scratch(device, n):
    """per-device scratch for the two-stage bias-gradient reduction (64 * out floats)"""
    key = str(device)
    t = _SCRATCH.get(key)
    if t is None or t.numel() < n:
        t = torch.empty(max(n, 64 * 1024), dtype=torch.float32, device=device)
        _SCRATCH[key] = t
    return t


_PACKED = {}  # W.data_ptr() -> (W._version, forward-packed view, dgrad-packed view, arena, W)
# (the entry keeps W alive so that its address cannot be recycled by another tensor with an equal version)


def prepack(weights):
    """Pack every MFMA-shaped dense weight of a training step with ONE launch (forward and dgrad layouts);
    the optimizer rewrites all of them every step.  `Linear` picks the packed copies up by (pointer, version)."""
    lib = _lib.load()
    todo = []
    for W in weights:
        if W.dim() != 2 or not W.is_contiguous() or not lib.tsd_linear_packable(W.shape[1], W.shape[0]):
            continue
        hit = _PACKED.get(W.data_ptr())
        if hit is None or hit[0] != W._version or hit[4]._cdata != W._cdata:
            todo.append(W)
    if not todo:
        return
    dev = todo[0].device
    arena = torch.empty(2 * sum(W.numel() for W in todo), dtype=torch.float32, device=dev)
    n = 2 * len(todo)
    Ws, dst, od, idim, tr = (C.c_void_p * n)(), (C.c_void_p * n)(), (C.c_int32 * n)(), (C.c_int32 * n)(), (C.c_int32 * n)()
    off = 0
    for k, W in enumerate(todo):
        views = []
        for t in (0, 1):
            v = arena[off:off + W.numel()]
            off += W.numel()
            Ws[2 * k + t], dst[2 * k + t] = W.data_ptr(), v.data_ptr()
            od[2 * k + t], idim[2 * k + t], tr[2 * k + t] = W.shape[0], W.shape[1], t
            views.append(v)
        _PACKED[W.data_ptr()] = (W._version, views[0], views[1], arena, W)
    check(lib.tsd_pack_linear_batch(n, Ws, dst, od, idim, tr, stream_ptr()))
    if len(_PACKED) > 512:  # parameters of models that went away
        for key in [k for k, v in _PACKED.items() if v[3] is not arena][:256]:
            del _PACKED[key]


def _packed(W, which):
    hit = _PACKED.get(W.data_ptr())
    return hit[which] if hit is not None and hit[0] == W._version and hit[4]._cdata == W._cdata else None


class Linear(torch.autograd.Function):
    """y = x W^T + b   (torch.nn.Linear semantics; W [out,in])"""

    @staticmethod
    def forward(ctx, x, W, b):
        lib = _lib.load()
        x, W = _c(x), _c(W)
        rows, fin = x.shape
        out = W.shape[0]
        y = torch.empty(rows, out, dtype=torch.float32, device=x.device)
        wp = _packed(W, 1)
        if wp is not None:
            check(lib.tsd_linear_fwd_packed(rows, fin, out, ptr(x), ptr(wp), ptr(b), ptr(y), stream_ptr()))
        else:
            sc = _scratch(x.device, fin * out)
            check(lib.tsd_linear_fwd(rows, fin, out, ptr(x), ptr(W), ptr(b), ptr(y), ptr(sc), sc.numel(),
                                     stream_ptr()))
        ctx.save_for_backward(x, W)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, W = ctx.saved_tensors
        dy = _c(dy)
        rows, fin = x.shape
        out = W.shape[0]
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dW = torch.empty_like(W) if ctx.needs_input_grad[1] else None
        db = torch.empty(out, dtype=torch.float32, device=x.device) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        sc = _scratch(x.device, 64 * out + fin * out + 64 * out * fin)  # bias partials | packed W | wgrad partials
        check(lib.tsd_linear_bwd(rows, fin, out, ptr(x), ptr(W), ptr(_packed(W, 2)), ptr(dy), ptr(dx), ptr(dW), ptr(db),
                                 ptr(sc), sc.numel(), stream_ptr()))
        return dx, dW, db


class Act(torch.autograd.Function):
    """kind 0: swish, 1: shifted softplus"""

    @staticmethod
    def forward(ctx, x, kind):
        lib = _lib.load()
        x = _c(x)
        y = torch.empty_like(x)
        check(lib.tsd_act_fwd(kind, x.numel(), ptr(x), ptr(y), stream_ptr()))
        ctx.save_for_backward(x)
        ctx.kind = kind
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        (x,) = ctx.saved_tensors
        dy = _c(dy)
        dx = torch.empty_like(x)
        check(lib.tsd_act_bwd(ctx.kind, x.numel(), ptr(x), ptr(dy), ptr(dx), stream_ptr()))
        return dx, None


class EmbMul(torch.autograd.Function):
    """y = x * emb[idx]   (idx uint8 edge types)"""

    @staticmethod
    def forward(ctx, x, emb, idx):
        lib = _lib.load()
        x, emb = _c(x), _c(emb)
        rows, H = x.shape
        y = torch.empty_like(x)
        check(lib.tsd_emb_mul_fwd(rows, H, ptr(x), ptr(emb), ptr(idx), ptr(y), stream_ptr()))
        ctx.save_for_backward(x, emb, idx)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, emb, idx = ctx.saved_tensors
        dy = _c(dy)
        rows, H = x.shape
        dx = torch.empty_like(x)
        demb = torch.zeros_like(emb)
        check(lib.tsd_emb_mul_bwd(rows, H, ptr(x), ptr(emb), ptr(idx), ptr(dy), ptr(dx), ptr(demb), stream_ptr()))
        return dx, demb, None


class GatherRows(torch.autograd.Function):
    """y = table[idx]   (idx int64)"""

    @staticmethod
    def forward(ctx, table, idx):
        lib = _lib.load()
        table = _c(table)
        rows, H = idx.shape[0], table.shape[1]
        y = torch.empty(rows, H, dtype=torch.float32, device=table.device)
        check(lib.tsd_gather_rows(rows, H, ptr(table), ptr(idx), ptr(y), stream_ptr()))
        ctx.save_for_backward(idx)
        ctx.shape = table.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        (idx,) = ctx.saved_tensors
        dy = _c(dy)
        dt = torch.zeros(ctx.shape, dtype=torch.float32, device=dy.device)
        check(lib.tsd_scatter_rows_add(idx.shape[0], ctx.shape[1], ptr(dy), ptr(idx), ptr(dt), stream_ptr()))
        return dt, None


class RowMask(torch.autograd.Function):
    """y = x * C(dist)[:, None]   (CFConv cutoff weight: hard mask or cosine, schnet.py:92-99)"""

    @staticmethod
    def forward(ctx, x, dist, cutoff, smooth):
        lib = _lib.load()
        y = x.clone(memory_format=torch.contiguous_format)
        check(lib.tsd_row_mask(y.shape[0], y.shape[1], ptr(dist), float(cutoff), int(smooth), ptr(y), stream_ptr()))
        ctx.save_for_backward(dist)
        ctx.cutoff, ctx.smooth = float(cutoff), int(smooth)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        (dist,) = ctx.saved_tensors
        dx = dy.clone(memory_format=torch.contiguous_format)
        check(lib.tsd_row_mask(dx.shape[0], dx.shape[1], ptr(dist), ctx.cutoff, ctx.smooth, ptr(dx), stream_ptr()))
        return dx, None, None, None


class Aggregate(torch.autograd.Function):
    """agg[i] = sum_{e in row i} x1[dst e] * Wf[umap e]   (schnet.py:102-107)"""

    @staticmethod
    def forward(ctx, x1, Wf, db):
        lib = _lib.load()
        x1, Wf = _c(x1), _c(Wf)
        N, H = x1.shape
        out = torch.empty_like(x1)
        check(lib.tsd_cfconv_aggregate(H, N, ptr(db.enc.row_ptr), ptr(db.enc.dst), ptr(db.enc.umap), ptr(Wf),
                                       ptr(x1), ptr(out), stream_ptr()))
        ctx.save_for_backward(x1, Wf)
        ctx.db = db
        return out

    @staticmethod
    def backward(ctx, dagg):
        lib = _lib.load()
        x1, Wf = ctx.saved_tensors
        db = ctx.db
        dagg = _c(dagg)
        N, H = x1.shape
        # the edge set and Wf are symmetric: the adjoint w.r.t. x1 is the same row-wise aggregation of dagg
        dx1 = torch.empty_like(x1)
        check(lib.tsd_cfconv_aggregate(H, N, ptr(db.enc.row_ptr), ptr(db.enc.dst), ptr(db.enc.umap), ptr(Wf),
                                       ptr(dagg), ptr(dx1), stream_ptr()))
        dWf = torch.empty_like(Wf)
        check(lib.tsd_aggregate_bwd_filter(H, Wf.shape[0], db.enc_u.struct(), ptr(dagg), ptr(x1), ptr(dWf),
                                           stream_ptr()))
        return dx1, dWf, None


class PairProduct(torch.autograd.Function):
    """p[u] = h[i] * h[j] over the undirected out edges"""

    @staticmethod
    def forward(ctx, h, db, Eo):
        lib = _lib.load()
        h = _c(h)
        H = h.shape[1]
        p = torch.empty(Eo, H, dtype=torch.float32, device=h.device)
        check(lib.tsd_pair_product_fwd(H, Eo, db.out_u.struct(), ptr(h), ptr(p), stream_ptr()))
        ctx.save_for_backward(h)
        ctx.db = db
        return p

    @staticmethod
    def backward(ctx, dp):
        lib = _lib.load()
        (h,) = ctx.saved_tensors
        dp = _c(dp)
        dh = torch.empty_like(h)
        check(lib.tsd_pair_product_bwd(h.shape[0], h.shape[1], ctx.db.out.struct(), ptr(dp), ptr(h), ptr(dh),
                                       stream_ptr()))
        return dh, None, None


class EqUndirected(torch.autograd.Function):
    """eq_transform (geometry.py:22-30) of a per-undirected-pair score; differentiable w.r.t. the score"""

    @staticmethod
    def forward(ctx, s_u, pos, db):
        lib = _lib.load()
        s_u, pos = _c(s_u), _c(pos)
        out = torch.empty(pos.shape[0], 3, dtype=torch.float32, device=pos.device)
        check(lib.tsd_eq_und_fwd(pos.shape[0], db.out.struct(), ptr(pos), ptr(s_u), ptr(out), stream_ptr()))
        ctx.save_for_backward(pos)
        ctx.db = db
        ctx.n = s_u.shape[0]
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        (pos,) = ctx.saved_tensors
        g = _c(g)
        ds = torch.empty(ctx.n, dtype=torch.float32, device=g.device)
        check(lib.tsd_eq_und_bwd(ctx.n, ctx.db.out_u.struct(), ptr(pos), ptr(g), ptr(ds), stream_ptr()))
        return ds, None, None


def pair_distance(db, which, pos, E):
    lib = _lib.load()
    lst = db.out_u if which == "out_u" else db.enc_u
    d = torch.empty(E, dtype=torch.float32, device=pos.device)
    check(lib.tsd_pair_distance(E, lst.struct(), ptr(_c(pos)), ptr(d), stream_ptr()))
    return d


def linear(x, W, b=None):
    return Linear.apply(x, W, b)


def swish(x):
    return Act.apply(x, 0)


def ssp(x):
    return Act.apply(x, 1)


def train_forward(model, db, pos):
    """The reference forward_ (condensenc.py:178-239) on the current geometry, differentiable w.r.t. the
    parameters.  Returns (s_u [Eo] edge_inv per undirected out edge, Eo)."""
    P = dict(model.named_parameters())
    cfg = model._cfg
    L = cfg.num_convs
    prepack(P.values())
    db.geometry(pos)
    Eu, Eo = db.enc_u.num_edges(), db.out_u.num_edges()  # host syncs (training is not latency critical)

    a = GatherRows.apply(P["atom_embedding.weight"], db.atom_type)
    wf = P["atom_feat_embedding.weight"]
    fr = linear(db.r_feat.to(torch.float32), wf)
    fp = linear(db.p_feat.to(torch.float32), wf)
    h = torch.cat([a + fr, fp - fr], dim=-1)  # condensenc.py:196-198

    def embed(lst, E):  # condensenc.py:156-176 'bond_w_d', edge.py:58-68
        d = lst.dist[:E].clone().unsqueeze(-1)
        e = linear(swish(linear(d, P["edge_encoder.mlp.layers.0.weight"], P["edge_encoder.mlp.layers.0.bias"])),
                   P["edge_encoder.mlp.layers.1.weight"], P["edge_encoder.mlp.layers.1.bias"])
        emb = P["edge_encoder.bond_emb.weight"]
        c = torch.cat([EmbMul.apply(e, emb, lst.type_r[:E]), EmbMul.apply(e, emb, lst.type_p[:E])], dim=-1)
        s1 = swish(linear(c, P["edge_cat.0.weight"], P["edge_cat.0.bias"]))
        return linear(s1, P["edge_cat.2.weight"], P["edge_cat.2.bias"])

    ea = embed(db.enc_u, Eu)
    dist_u = db.enc_u.dist[:Eu].clone()
    for l in range(L):  # schnet.py:88-128, 223-224
        p = f"encoder.interactions.{l}."
        Wf = linear(ssp(linear(ea, P[p + "conv.nn.0.weight"], P[p + "conv.nn.0.bias"])),
                    P[p + "conv.nn.2.weight"], P[p + "conv.nn.2.bias"])
        Wf = RowMask.apply(Wf, dist_u, cfg.conv_cutoff, cfg.smooth_conv)
        x1 = linear(h, P[p + "conv.lin1.weight"])
        agg = Aggregate.apply(x1, Wf, db)
        x = linear(agg, P[p + "conv.lin2.weight"], P[p + "conv.lin2.bias"])
        h = h + linear(ssp(x), P[p + "lin.weight"], P[p + "lin.bias"])
    ea_o = embed(db.out_u, Eo)
    hp = torch.cat([PairProduct.apply(h, db, Eo), ea_o], dim=-1)  # common.py:226-229
    g = "grad_dist_mlp.layers."
    x = swish(linear(hp, P[g + "0.weight"], P[g + "0.bias"]))
    x = swish(linear(x, P[g + "1.weight"], P[g + "1.bias"]))
    s_u = linear(x, P[g + "2.weight"], P[g + "2.bias"]).view(-1)
    return s_u, Eo


# ---------------------------------------------------------------------------------------------
# dual-encoder (GeoDiff legacy) network, reference models/epsnet/dualenc.py:206-374
class Gine(torch.autograd.Function):
    """GINEConv message pass + self term over the local edges of the symmetric extended graph
    (gin.py:61-73), edge attributes once per undirected pair."""

    @staticmethod
    def forward(ctx, x, ea_u, db, act, eps):
        lib = _lib.load()
        x, ea_u = _c(x), _c(ea_u)
        N, H = x.shape
        out = torch.empty_like(x)
        check(lib.tsd_gine_csr_fwd(N, H, act, eps, db.enc.struct(), ptr(ea_u), ptr(x), ptr(out), stream_ptr()))
        ctx.save_for_backward(x, ea_u)
        ctx.db, ctx.act, ctx.eps = db, act, eps
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        x, ea_u = ctx.saved_tensors
        dout = _c(dout)
        N, H = x.shape
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dea = torch.empty_like(ea_u) if ctx.needs_input_grad[1] else None
        check(lib.tsd_gine_csr_bwd(N, ea_u.shape[0], H, ctx.act, ctx.eps, ctx.db.enc.struct(), ctx.db.enc_u.struct(),
                                   ptr(ea_u), ptr(x), ptr(dout), ptr(dx), ptr(dea), stream_ptr()))
        return dx, dea, None, None, None


_ACT_KIND = {"swish": 0, "ReLU": 2, "Softplus": 3}
_GINE_ACT = {None: 0, "ReLU": 1, "Softplus": 2}


def dual_forward(model, db, pos):
    """Both heads of DualEncoderEpsNetwork.forward on the current geometry, per UNDIRECTED pair of the
    extended graph (every per-edge quantity of the reference is symmetric under i <-> j).  Differentiable
    w.r.t. the parameters when autograd is on."""
    lib = _lib.load()
    P = dict(model.named_parameters())
    cfg = db.cfg
    H = cfg.hidden
    prepack(P.values())
    db.geometry(pos)
    Eu = db.enc_u.num_edges()  # host sync; the edge set is symmetric: E = 2 Eu
    E = 2 * Eu
    t_u, t_dir = db.enc_u.type_r[:Eu], db.enc.type_r[:E]
    if model.TS:  # dualenc.py:266-283: bonded pairs carry (r, p), k-hop pairs 22 + k - 1 in both slots
        def raw(lst, n):
            code = db.pair_code_raw[lst.pair_id[:n].long()].to(torch.int32) & 0xFFFF
            return (code & 31), ((code >> 5) & 31)
        r_u, p_u = raw(db.enc_u, Eu)
        bonded_u = t_u == 1
        t1_u = torch.where(bonded_u, r_u.to(torch.uint8), t_u).contiguous()
        t2_u = torch.where(bonded_u, p_u.to(torch.uint8), t_u).contiguous()
        r_d, p_d = raw(db.enc, E)
        edge_type_dir = torch.where(t_dir == 1, (r_d * 22 + p_d).to(torch.int64), t_dir.to(torch.int64) + 462)
    else:
        t1_u = t2_u = t_u
        edge_type_dir = torch.where(t_dir < 22, t_dir.to(torch.int64), t_dir.to(torch.int64) + 462)
    edge_type_dir = torch.where(t_dir == 0, torch.zeros_like(edge_type_dir), edge_type_dir)

    act = _ACT_KIND[model.mlp_act]
    d = db.enc_u.dist[:Eu].clone().unsqueeze(-1)
    dist_u = db.enc_u.dist[:Eu].clone()

    def embed(enc, cat):  # edge.py:58-68 (+ dualenc.py:266-283 for TS)
        e = linear(Act.apply(linear(d, P[enc + ".mlp.layers.0.weight"], P[enc + ".mlp.layers.0.bias"]), act),
                   P[enc + ".mlp.layers.1.weight"], P[enc + ".mlp.layers.1.bias"])
        emb = P[enc + ".bond_emb.weight"]
        if not model.TS:
            return EmbMul.apply(e, emb, t1_u)
        c = torch.cat([EmbMul.apply(e, emb, t1_u), EmbMul.apply(e, emb, t2_u)], dim=-1)
        s1 = Act.apply(linear(c, P[cat + ".0.weight"], P[cat + ".0.bias"]), _ACT_KIND[model.edge_cat_act])
        return linear(s1, P[cat + ".2.weight"], P[cat + ".2.bias"])

    def pair_mlp(prefix, hp):  # common.py:226-229 + MultiLayerPerceptron
        x = Act.apply(linear(hp, P[prefix + ".layers.0.weight"], P[prefix + ".layers.0.bias"]), act)
        x = Act.apply(linear(x, P[prefix + ".layers.1.weight"], P[prefix + ".layers.1.bias"]), act)
        return linear(x, P[prefix + ".layers.2.weight"], P[prefix + ".layers.2.bias"]).view(-1)

    # ---- global head: SchNet with its own node embedding (schnet.py:203-225, embed_node=True) ----
    ea_g = embed("edge_encoder_global", "edge_cat_global")
    w = P["encoder_global.node_emb.weight"]
    with torch.no_grad():  # nn.Embedding(max_norm=10) renormalises the looked-up rows in place
        check(lib.tsd_embedding_renorm(w.shape[0], H, db.N, ptr(db.atom_type), 10.0, ptr(w), ptr(db.renorm_scratch),
                                       stream_ptr()))
    h = GatherRows.apply(w, db.atom_type)
    for l in range(cfg.num_convs):
        p = f"encoder_global.interactions.{l}."
        Wf = linear(ssp(linear(ea_g, P[p + "conv.nn.0.weight"], P[p + "conv.nn.0.bias"])),
                    P[p + "conv.nn.2.weight"], P[p + "conv.nn.2.bias"])
        Wf = RowMask.apply(Wf, dist_u, cfg.conv_cutoff, cfg.smooth_conv)
        x1 = linear(h, P[p + "conv.lin1.weight"])
        agg = Aggregate.apply(x1, Wf, db)
        x = linear(agg, P[p + "conv.lin2.weight"], P[p + "conv.lin2.bias"])
        h = h + linear(ssp(x), P[p + "lin.weight"], P[p + "lin.bias"])
    s_global_u = pair_mlp("grad_global_dist_mlp", torch.cat([PairProduct.apply(h, db, Eu), ea_g], dim=-1))

    # ---- local head: GINE on the edges with type > 0 (gin.py:79-149) ----
    ea_l = embed("edge_encoder_local", "edge_cat_local")
    enc = model.encoder_local
    gact, nact = _GINE_ACT[enc.activation], (_ACT_KIND[enc.activation] if enc.activation else None)
    x = GatherRows.apply(P["encoder_local.node_emb.weight"], db.atom_type)
    hiddens = []
    for k, conv in enumerate(enc.convs):
        q = f"encoder_local.convs.{k}.nn.layers."
        agg = Gine.apply(x, ea_l, db, gact, float(conv.initial_eps))
        y = linear(agg, P[q + "0.weight"], P[q + "0.bias"])
        if nact is not None:
            y = Act.apply(y, nact)
        y = linear(y, P[q + "1.weight"], P[q + "1.bias"])
        if k < len(enc.convs) - 1 and nact is not None:
            y = Act.apply(y, nact)
        if enc.short_cut:
            y = y + x
        hiddens.append(y)
        x = y
    hl = torch.cat(hiddens, dim=-1) if enc.concat_hidden else hiddens[-1]
    s_local_u = pair_mlp("grad_local_dist_mlp", torch.cat([PairProduct.apply(hl, db, Eu), ea_l], dim=-1))
    return {"s_global_u": s_global_u, "s_local_u": s_local_u, "Eu": Eu, "E": E, "emb_type_u": t_u,
            "emb_type_dir": t_dir, "edge_type_dir": edge_type_dir}


# ---------------------------------------------------------------------------------------------
# The training step as two library calls (tsdiff_amd/csrc/train_step.hip): the whole forward + loss and the
# whole backward are sequenced in C++; autograd sees ONE node whose inputs are the parameters.
class FusedTrainLoss(torch.autograd.Function):
    """loss (N,1) of CondenseEncoderEpsNetwork.get_loss (condensenc.py:267-328), differentiable w.r.t. every
    parameter.  `params` come in the order of engine.raw_param_names."""

    @staticmethod
    def forward(ctx, model, db, pos0, pos_perturbed, a_graph, *params):
        lib = _lib.load()
        cfg = model._cfg
        dev = pos0.device
        # parameters re-homed in one flat buffer (optim.flatten_parameters) are handed over as they lie
        # (EVERY parameter must still be the view it was made: a re-homed one -- p.data = ..., weight tying, a partial
        # .to() -- would otherwise train on the stale copy in the flat buffer)
        from . import optim
        flat = getattr(model, "_flat_param", None)
        live = flat is not None and flat.device == dev and optim._views_of(params, flat)
        if live:
            raw = flat.detach()
        else:
            raw = torch.cat([p.detach().reshape(-1) for p in params])
        n_raw = lib.tsd_train_raw_floats(C.byref(cfg))
        if raw.numel() != n_raw:
            raise ValueError(f"parameters hold {raw.numel()} floats, the config needs {n_raw}")
        nws = lib.tsd_train_workspace_floats(C.byref(cfg), db.N, db.P)
        ws = getattr(model, "_train_ws", None)  # one arena per model, grown on demand, reused by every batch
        if ws is None or ws.numel() < nws or ws.device != dev:
            ws = None
            model._train_ws = None  # release before growing
            ws = torch.empty(max(int(nws * 1.15), 1), dtype=torch.float32, device=dev)
            model._train_ws = ws
        h2 = OPTIONS.train_gemm == "h2" and not getattr(model, "_train_f32", False)
        b = db.train_struct(h2)
        counts = (C.c_int32 * 4)()  # undirected edge counts of the two lists, topology status word, separately embedded out edges
        loss = torch.empty(db.N, 1, dtype=torch.float32, device=dev)
        pos0, pos_perturbed, a_graph = _c(pos0.float()), _c(pos_perturbed.float()), _c(a_graph.float())
        # a batch prefetched with its positions (CondenseEncoderEpsNetwork.prefetch_batch(pos=...)): the perturbed geometry's
        # edge lists were built ahead on the side stream and their counts are in pinned memory behind an event that fired
        # long ago -- the library neither rebuilds them nor waits for the device (tsd_batch.reserved bit 7)
        stash = getattr(db, "train_stash", None)
        db.train_stash = None  # (one-shot: any other geometry build on this batch overwrites the lists)
        if stash is not None and stash["pos_perturbed"] is pos_perturbed:
            stash["event"].synchronize()
            for k in range(4):
                counts[k] = int(stash["counts"][k])
            b.reserved |= 128
        check(lib.tsd_train_forward(C.byref(cfg), C.byref(b), ptr(raw), ptr(db.atom_type), ptr(db.r_feat),
                                    ptr(db.p_feat), ptr(pos0), ptr(pos_perturbed), ptr(a_graph), ptr(db.status),
                                    ptr(ws), ws.numel(), ptr(loss), counts, stream_ptr()))
        db.check_status(counts[2])  # malformed bond lists raise ValueError here (results are discarded)
        # The saved activations live in the model's shared arena and the edge lists in the cached batch: both are
        # overwritten by the next get_loss / forward / geometry build.  Stamp them so that backward can tell.
        db.geo_gen += 1
        model._train_ws_gen = getattr(model, "_train_ws_gen", 0) + 1
        ctx.stamp = (db.geo_gen, model._train_ws_gen)
        # the live flat buffer is read again by backward (weights of the dgrads): an optimizer step or any other
        # in-place update of a parameter in between would differentiate against other weights than the forward used
        # (flat form: `params` is the one flat leaf; the versions that move are the real Parameters')
        real = getattr(model, "_flat_real_params", None) if (len(params) == 1 and params[0] is getattr(model, "_flat_leaf", None)) else None
        ctx.flat_form = real is not None
        vparams = real if real is not None else params
        ctx.param_versions = (flat._version + sum(p._version for p in vparams)) if live else None
        ctx.params = vparams if live else None
        ctx.used = False
        ctx.model, ctx.db, ctx.raw, ctx.ws, ctx.counts, ctx.pos = model, db, raw, ws, counts, pos_perturbed
        ctx.sizes = [p.numel() for p in params]
        ctx.shapes = [p.shape for p in params]
        ctx.h2 = h2
        if h2:  # what a recomputation of this forward in fp32 needs (backward: range flag)
            ctx.fwd_in, ctx.loss = (pos0, a_graph), loss
        return loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dloss):
        lib = _lib.load()
        cfg, db = ctx.model._cfg, ctx.db
        if ctx.used:
            raise RuntimeError("tsdiff_amd: the fused training step supports ONE backward per get_loss "
                               "(tsd_train_backward consumes its saved activations in place)")
        if ctx.stamp != (db.geo_gen, getattr(ctx.model, "_train_ws_gen", 0)):
            raise RuntimeError(
                "tsdiff_amd: the activations of this get_loss were overwritten before backward() -- another "
                "get_loss / forward / sampling call ran on the same model or batch in between.  The fused training "
                "step keeps ONE step's activations per model (reference order: get_loss, backward, step; "
                "train.py:128-145).  Call backward first, or set TSDIFF_TRAIN=ops for the op-by-op autograd form.")
        if ctx.param_versions is not None and \
                ctx.param_versions != ctx.model._flat_param._version + sum(p._version for p in ctx.params):
            raise RuntimeError(
                "tsdiff_amd: a parameter was modified in place between get_loss and backward() (an optimizer step of "
                "another loss?): the fused training step differentiates against the live flat parameter buffer.  "
                "Call backward before stepping, or set TSDIFF_TRAIN=ops.")
        ctx.used = True
        if ctx.flat_form:
            # ONE persistent flat gradient per model: its per-parameter views are made once and handed out again every step
            # (80 views + 80 AccumulateGrad nodes per step were ~0.5 ms of a 2-ms loop that is as long as its host side)
            grad = getattr(ctx.model, "_flat_grad_buf", None)
            if grad is None or grad.shape != ctx.raw.shape or grad.device != ctx.raw.device:
                grad = ctx.model._flat_grad_buf = torch.empty_like(ctx.raw)
                ctx.model._flat_grad_views = None
        else:
            grad = torch.empty_like(ctx.raw)
        dloss = _c(dloss.float()).view(-1)
        # data-parallel step (distributed.dp_backward set `_dp_early_reduce`): the interaction blocks' gradients -- one
        # contiguous range, 83 % of the vector -- are final long before the call's last kernel; the library records an
        # event there and their all-reduce starts on a side stream beside the embedding's backward chain
        early = getattr(ctx.model, "_dp_early_reduce", None)
        ev = None
        if early is not None:
            ev = torch.cuda.Event()
            ev.record()  # (creates the hipEvent_t; the library records it again where the bucket is complete)

        def run(h2):
            b = db.train_struct(h2)
            return lib.tsd_train_backward2(C.byref(cfg), C.byref(b), ptr(ctx.raw), ptr(db.atom_type), ptr(ctx.pos),
                                           ptr(ctx.ws), ctx.ws.numel(), ctx.counts, ptr(dloss), ptr(grad),
                                           C.c_void_p(ev.cuda_event) if ev is not None else None, stream_ptr())
        code = run(ctx.h2)
        if code == _lib.TSD_ERR_RANGE:
            # an activation of the split-f16 forward left the f16 range (the library read the flag behind the forward's
            # last kernel and launched nothing): the forward is recomputed on the fp32-input MFMA kernels INTO THE SAME loss
            # tensor (scalars the caller derived from it before backward() are stale for this one step; the tensor itself
            # now holds the fp32 step's values) and the backward runs in fp32.  The fallback is PER STEP: the next
            # get_loss tries the split-f16 kernels again (one tiny aggregate at the cutoff edge must not change the
            # arithmetic of a whole training run); `model._h2_range_trips` counts the trips, and only
            # OPTIONS.train_fallback_latch CONSECUTIVE ones (a checkpoint whose activations really live beyond 65504)
            # latch the model to fp32, because every trip costs a second forward.
            import warnings
            m = ctx.model
            m._h2_range_trips = getattr(m, "_h2_range_trips", 0) + 1
            m._h2_range_run = getattr(m, "_h2_range_run", 0) + 1
            latch = m._h2_range_run >= OPTIONS.train_fallback_latch
            warnings.warn("tsdiff_amd: an activation left the range the split-f16 training arithmetic covers; this step was "
                          "recomputed on the fp32-input MFMA kernels (trip %d%s)" % (
                              m._h2_range_trips, "; %d in a row: the model now trains in fp32" % m._h2_range_run if latch else ""),
                          RuntimeWarning, stacklevel=2)
            if latch:
                m._train_f32 = True
            db.status[:1].zero_()
            pos0, a_graph = ctx.fwd_in
            check(lib.tsd_train_forward(C.byref(cfg), C.byref(db.train_struct(False)), ptr(ctx.raw), ptr(db.atom_type),
                                        ptr(db.r_feat), ptr(db.p_feat), ptr(pos0), ptr(ctx.pos), ptr(a_graph), None,
                                        ptr(ctx.ws), ctx.ws.numel(), ptr(ctx.loss), ctx.counts, stream_ptr()))
            code = run(False)
        elif ctx.h2:
            ctx.model._h2_range_run = 0
        check(code)
        ctx.model._dp_early_done = None
        if early is not None:
            lay = (C.c_size_t * 3)()
            check(lib.tsd_train_grad_buckets(C.byref(cfg), lay))
            off, cnt = int(lay[0]), int(lay[1])
            from .engine import _side_stream
            side = _side_stream(grad.device)
            side.wait_event(ev)
            with torch.cuda.stream(side):
                early(grad[off:off + cnt])
            grad.record_stream(side)
            ctx.model._dp_early_done = (side, off, cnt)
        ctx.model._flat_grad = grad  # (distributed.dp_backward all-reduces it in place)
        if ctx.flat_form:
            real = ctx.params
            views = ctx.model._flat_grad_views
            if views is None:
                views = ctx.model._flat_grad_views = [g.view(p.shape) for g, p in zip(grad.split([p.numel() for p in real]), real)]
            if all(p.grad is None for p in real):
                for p, v in zip(real, views):
                    p.grad = v
            else:  # a .grad appeared between get_loss and backward: torch's accumulate semantics
                for p, v in zip(real, views):
                    p.grad = v.clone() if p.grad is None else p.grad + v
            return (None, None, None, None, None, None)
        views = [g.view(s) for g, s in zip(grad.split(ctx.sizes), ctx.shapes)]  # views of ONE flat buffer
        return (None, None, None, None, None) + tuple(views)


def fused_train_loss(model, db, pos0, pos_perturbed, a_graph):
    """The flat form (round 6): when the parameters are views of ONE flat buffer (optim.flatten_parameters: what
    optim.get_optimizer sets up) and none of them carries a .grad, autograd sees ONE leaf -- a tensor that shares the flat
    buffer's storage -- instead of ~80 parameters; the backward writes a persistent flat gradient and assigns its cached
    per-parameter views to the Parameters' .grad itself.  Same gradients, bit for bit; what differs from torch's default:
    the .grad tensors of consecutive steps are views of the SAME memory (as with zero_grad(set_to_none=False)), and
    torch.autograd.grad(loss, parameters) does not see the parameters (OPTIONS.train_flat_grad = False, or any existing
    .grad, selects the per-parameter form)."""
    params = model.raw_params()
    flat = getattr(model, "_flat_param", None)
    # (every parameter trainable and hook-free: a frozen parameter must not receive a .grad, tensor hooks and
    # post-accumulate-grad hooks only fire on the per-parameter path)
    if OPTIONS.train_flat_grad and flat is not None and flat.is_cuda and \
            all(p.grad is None and p.requires_grad and not p._backward_hooks and
                not getattr(p, "_post_accumulate_grad_hooks", None) for p in params):
        from . import optim
        cache = getattr(model, "_flat_leaf_key", None)
        key = (flat._cdata, tuple(id(p) for p in params))
        if cache != key:
            ok = optim._views_of(params, flat)
            model._flat_leaf_key = key
            model._flat_leaf = flat.detach().requires_grad_(True) if ok else None
            model._flat_real_params = list(params) if ok else None
            model._flat_grad_views = None
            offs, o = [], 0
            for p in params:
                offs.append(o * flat.element_size())
                o += p.numel()
            model._flat_offsets = offs
        if model._flat_leaf is not None:
            # every step: the Parameters must still be the views they were made (a re-homed one -- p.data = ..., weight
            # tying, a partial .to() -- would otherwise train on the stale copy in the flat buffer): one data_ptr() each
            base = flat.data_ptr()
            if all(p.data_ptr() == base + off for p, off in zip(params, model._flat_offsets)):
                return FusedTrainLoss.apply(model, db, pos0, pos_perturbed, a_graph, model._flat_leaf)
            model._flat_leaf_key = None  # (re-validated, or abandoned, at the next call)
    return FusedTrainLoss.apply(model, db, pos0, pos_perturbed, a_graph, *params)
