"""TEST INFRASTRUCTURE ONLY -- the CPU oracle for the TSDiff score-network hot path.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may
import this module; the product path (tsdiff_amd/) never does and fails loudly
when its HIP library is missing.

This is a restatement (our own code, torch-CPU, no PyG) of the reference
algorithm.  Each function cites the reference file:line it follows.  It is
PINNED: tests/test_oracle_golden.py checks it against golden vectors that
`oracle/gen_golden.py` produced by running the unchanged reference sources
(through the import shims of oracle/ref_shims.py) in the build container.

Differences from the reference, all result-preserving:
  * graph extension is done per graph (block diagonal) with BFS hop counts
    instead of dense (N,N) matrix powers over the whole batch
    (reference models/common.py:115-202) -- same edges, same types, same order;
  * `mlp(d)` of the edge encoder is evaluated once and shared by the r/p
    branches (the reference evaluates it twice with identical inputs,
    models/epsnet/condensenc.py:169-170).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

NUM_BOND_TYPES = 22  # reference utils/chem.py:21 with rdkit 2020.09 BondType.names


# ----------------------------------------------------------------------------
# schedule                                  reference condensenc.py:13-43,91-102
# ----------------------------------------------------------------------------
def beta_schedule(cfg):
    kind = cfg["beta_schedule"]
    T = int(cfg["num_diffusion_timesteps"])
    b0, b1 = float(cfg["beta_start"]), float(cfg["beta_end"])
    if kind == "sigmoid":
        x = np.linspace(-6, 6, T)
        betas = 1 / (np.exp(-x) + 1) * (b1 - b0) + b0
    elif kind == "linear":
        betas = np.linspace(b0, b1, T, dtype=np.float64)
    elif kind == "quad":
        betas = np.linspace(b0 ** 0.5, b1 ** 0.5, T, dtype=np.float64) ** 2
    elif kind == "const":
        betas = b1 * np.ones(T, dtype=np.float64)
    elif kind == "jsd":
        betas = 1.0 / np.linspace(T, 1, T, dtype=np.float64)
    else:
        raise NotImplementedError(kind)
    betas = torch.from_numpy(betas).float()
    alphas = (1.0 - betas).cumprod(dim=0)
    return betas, alphas


def swish(x):  # reference utils/activation_functions.py:10-11
    return x * x.sigmoid()


def ssp(x):  # reference models/encoder/schnet.py:65-71
    return F.softplus(x) - math.log(2.0)


# ----------------------------------------------------------------------------
# graph extension      reference common.py:115-223, 328-384; condensenc.py:117-154
# ----------------------------------------------------------------------------
def _hop_matrix(n, src, dst, max_order):
    """Directed shortest-path hop count, 0 where > max_order or unreachable
    (reference get_higher_order_adj_matrix, common.py:119-143)."""
    adj = np.zeros((n, n), dtype=bool)
    adj[src, dst] = True
    reach = np.eye(n, dtype=bool)
    step = adj | reach
    hop = np.zeros((n, n), dtype=np.int64)
    for k in range(1, max_order + 1):
        nxt = (reach.astype(np.int64) @ step.astype(np.int64)) > 0
        hop[nxt & ~reach] = k
        reach = nxt
    return hop


def pair_types(bond_index, bond_type, num_nodes_per_graph, order):
    """Per ordered intra-graph pair (i != j): (type_r, type_p) of the order-`order`
    extended TS graph (reference _extend_ts_graph_order, common.py:115-202).

    Returns dict graph -> (n, type_r (n,n), type_p (n,n)) with 0 = not a local edge
    in that graph, 1..21 bond, 22+h-1 for h-hop (h = 2..order)."""
    bi = np.asarray(bond_index)
    bt = np.asarray(bond_type)
    off = np.concatenate([[0], np.cumsum(num_nodes_per_graph)])
    out = []
    for g, n in enumerate(num_nodes_per_graph):
        lo, hi = off[g], off[g + 1]
        sel = (bi[0] >= lo) & (bi[0] < hi)
        s, d, t = bi[0, sel] - lo, bi[1, sel] - lo, bt[sel]
        res = []
        for comp in (t // NUM_BOND_TYPES, t % NUM_BOND_TYPES):
            m = comp != 0
            hop = _hop_matrix(n, s[m], d[m], order)
            tm = np.zeros((n, n), dtype=np.int64)
            np.add.at(tm, (s[m], d[m]), comp[m])  # to_dense_adj sums duplicates
            high = np.where(hop > 1, NUM_BOND_TYPES + hop - 1, 0)
            assert (tm * high == 0).all()  # common.py:168,182
            res.append(tm + high)
        out.append((int(n), res[0], res[1]))
    return out


def extend_graph(pos, bond_index, bond_type, num_nodes_per_graph, order, cutoff):
    """edge_index (2,E) row-major sorted, type_r (E,), type_p (E,)
    == reference CondenseEncoderEpsNetwork._extend_condensed_graph_edge
    (condensenc.py:117-154): local (<= order hops in R or P) pairs united with the
    radius graph (dist^2 < cutoff^2, same graph, no self loops, no neighbour cap)."""
    pos_np = np.asarray(pos, dtype=np.float32)
    off = np.concatenate([[0], np.cumsum(num_nodes_per_graph)])
    rows, cols, trs, tps = [], [], [], []
    c2 = np.float32(cutoff) * np.float32(cutoff)
    for g, (n, tr, tp) in enumerate(pair_types(bond_index, bond_type, num_nodes_per_graph, order)):
        p = pos_np[off[g]:off[g] + n]
        diff = p[:, None, :] - p[None, :, :]
        d2 = (diff * diff).sum(-1, dtype=np.float32)
        member = ((tr != 0) | (tp != 0) | (d2 < c2)) & ~np.eye(n, dtype=bool)
        r, c = np.nonzero(member)  # row-major
        rows.append(r + off[g])
        cols.append(c + off[g])
        trs.append(tr[r, c])
        tps.append(tp[r, c])
    ei = torch.from_numpy(np.stack([np.concatenate(rows), np.concatenate(cols)]).astype(np.int64))
    return ei, torch.from_numpy(np.concatenate(trs)), torch.from_numpy(np.concatenate(tps))


# ----------------------------------------------------------------------------
# network pieces
# ----------------------------------------------------------------------------
def get_distance(pos, edge_index):  # reference models/geometry.py:18-19
    return (pos[edge_index[0]] - pos[edge_index[1]]).norm(dim=-1)


def eq_transform(score_d, pos, edge_index, edge_length):  # reference geometry.py:22-30
    N = pos.size(0)
    dd_dr = (1.0 / edge_length) * (pos[edge_index[0]] - pos[edge_index[1]])
    a = torch.zeros(N, 3, dtype=pos.dtype).index_add_(0, edge_index[0], dd_dr * score_d)
    b = torch.zeros(N, 3, dtype=pos.dtype).index_add_(0, edge_index[1], -dd_dr * score_d)
    return a + b


def node_embedding(sd, atom_type, r_feat, p_feat):  # reference condensenc.py:193-198
    dt = sd["atom_embedding.weight"].dtype
    a = sd["atom_embedding.weight"][atom_type]
    wf = sd["atom_feat_embedding.weight"]
    fr = F.linear(r_feat.to(dt), wf)
    fp = F.linear(p_feat.to(dt), wf)
    return torch.cat([a + fr, fp - fr], dim=-1)


def edge_embedding(sd, edge_length, type_r, type_p):
    """reference condensenc.py:156-176 ('bond_w_d'), edge.py:58-68, common.py:74-90."""
    h = F.linear(edge_length, sd["edge_encoder.mlp.layers.0.weight"], sd["edge_encoder.mlp.layers.0.bias"])
    d_emb = F.linear(swish(h), sd["edge_encoder.mlp.layers.1.weight"], sd["edge_encoder.mlp.layers.1.bias"])
    emb = sd["edge_encoder.bond_emb.weight"]
    cat = torch.cat([d_emb * emb[type_r], d_emb * emb[type_p]], dim=-1)
    x = swish(F.linear(cat, sd["edge_cat.0.weight"], sd["edge_cat.0.bias"]))
    return F.linear(x, sd["edge_cat.2.weight"], sd["edge_cat.2.bias"])


def cfconv_aggregate(x1, W, edge_index, N):
    """T5: agg[i] = sum_{e: edge_index[1][e] == i} x1[edge_index[0][e]] * W[e]
    (reference schnet.py:102,106 via MessagePassing aggr='add')."""
    msg = x1[edge_index[0]] * W
    return torch.zeros(N, W.size(1), dtype=W.dtype).index_add_(0, edge_index[1], msg)


def schnet_encoder(sd, z, edge_index, edge_length, edge_attr, num_convs, cutoff, trace=None, smooth=False,
                   prefix="encoder."):
    """reference schnet.py:203-225 / 110-128 / 74-107."""
    h = z
    if smooth:  # schnet.py:92-96
        C = 0.5 * (torch.cos(edge_length * math.pi / cutoff) + 1.0)
        C = (C * (edge_length <= cutoff) * (edge_length >= 0.0)).to(z.dtype).view(-1, 1)
    else:
        C = (edge_length <= cutoff).to(z.dtype).view(-1, 1)
    for l in range(num_convs):
        p = f"{prefix}interactions.{l}."
        W = F.linear(ssp(F.linear(edge_attr, sd[p + "conv.nn.0.weight"], sd[p + "conv.nn.0.bias"])),
                     sd[p + "conv.nn.2.weight"], sd[p + "conv.nn.2.bias"]) * C
        x1 = F.linear(h, sd[p + "conv.lin1.weight"])
        agg = cfconv_aggregate(x1, W, edge_index, h.size(0))
        x = F.linear(agg, sd[p + "conv.lin2.weight"], sd[p + "conv.lin2.bias"])
        x = F.linear(ssp(x), sd[p + "lin.weight"], sd[p + "lin.bias"])
        h = h + x
        if trace is not None:
            if l == 0:
                trace["W0"], trace["x1_0"], trace["agg0"] = W, x1, agg
            trace[f"h{l + 1}"] = h
    return h


def pair_output(sd, h, edge_index, edge_attr):
    """reference common.py:226-229 + condensenc.py:236-237."""
    hp = torch.cat([h[edge_index[0]] * h[edge_index[1]], edge_attr], dim=-1)
    x = swish(F.linear(hp, sd["grad_dist_mlp.layers.0.weight"], sd["grad_dist_mlp.layers.0.bias"]))
    x = swish(F.linear(x, sd["grad_dist_mlp.layers.1.weight"], sd["grad_dist_mlp.layers.1.bias"]))
    return F.linear(x, sd["grad_dist_mlp.layers.2.weight"], sd["grad_dist_mlp.layers.2.bias"])


def forward(sd, cfg, atom_type, r_feat, p_feat, pos, bond_index, bond_type,
            num_nodes_per_graph, trace=None):
    """reference CondenseEncoderEpsNetwork.forward_ (condensenc.py:178-239).
    Returns (edge_inv (E,1), edge_index (2,E), edge_length (E,1))."""
    enc = cfg["encoder"]
    z = node_embedding(sd, atom_type, r_feat, p_feat)
    ei, tr, tp = extend_graph(pos, bond_index, bond_type, num_nodes_per_graph,
                              int(cfg["edge_order"]), float(cfg["edge_cutoff"]))
    d = get_distance(pos, ei).unsqueeze(-1)
    ea = edge_embedding(sd, d, tr, tp)
    if trace is not None:
        trace.update(z=z, enc_edge_index=ei, enc_type_r=tr, enc_type_p=tp, enc_edge_length=d, enc_edge_attr=ea)
    h = schnet_encoder(sd, z, ei, d, ea, int(enc["num_convs"]), float(enc["cutoff"]), trace,
                       smooth=bool(enc.get("smooth_conv", False)))
    if int(cfg["edge_order"]) != int(cfg["pred_edge_order"]):
        ei, tr, tp = extend_graph(pos, bond_index, bond_type, num_nodes_per_graph,
                                  int(cfg["pred_edge_order"]), float(cfg["edge_cutoff"]))
        d = get_distance(pos, ei).unsqueeze(-1)
        ea = edge_embedding(sd, d, tr, tp)
    if trace is not None:
        trace.update(out_type_r=tr, out_type_p=tp, out_edge_attr=ea)
    edge_inv = pair_output(sd, h, ei, ea)
    return edge_inv, ei, d


def ensemble_forward(sds, cfg, *args, **kw):
    """reference EnsembleSampler.forward (sampler.py:58-116): in-place sum, then /M."""
    edge_inv, ei, d = forward(sds[0], cfg, *args, **kw)
    for sd in sds[1:]:
        edge_inv = edge_inv + forward(sd, cfg, *args, **kw)[0]
    return edge_inv / len(sds), ei, d


# ----------------------------------------------------------------------------
# sampler                                      reference models/sampler.py:118-268
# ----------------------------------------------------------------------------
def clip_norm(vec, limit):  # sampler.py:265-268
    norm = torch.norm(vec, dim=-1, p=2, keepdim=True)
    denom = torch.where(norm > limit, limit / norm, torch.ones_like(norm))
    return vec * denom


def center_pos(pos, batch, num_graphs):  # sampler.py:260-262
    s = torch.zeros(num_graphs, 3, dtype=pos.dtype).index_add_(0, batch, pos)
    cnt = torch.zeros(num_graphs, dtype=pos.dtype).index_add_(0, batch, torch.ones_like(batch, dtype=pos.dtype))
    return pos - (s / cnt.clamp(min=1).unsqueeze(-1))[batch]


def sigmas_from_alphas(alphas):  # sampler.py:143
    return (1.0 - alphas).sqrt() / alphas.sqrt()


def _compute_alpha(betas, t):  # sampler.py:138-141
    beta = torch.cat([torch.zeros(1), betas], dim=0)
    return (1 - beta).cumprod(dim=0).index_select(0, t + 1)


def sample(sds, cfg, atom_type, r_feat, p_feat, pos_init, bond_index, bond_type, batch,
           num_nodes_per_graph, noises, n_steps, step_lr=1e-7, clip=1000.0, clip_pos=None,
           sampling_type="ld", denoise_from_time_t=None, noise_from_time_t=None, init_noise=None):
    """reference EnsembleSampler.dynamic_sampling default branch (sampler.py:179-254)
    with the per-step Gaussian noise INJECTED (`noises[k]` is the k-th draw of
    `torch.randn_like(pos)`), so that a device implementation can be compared
    element-wise.  Returns (pos, [pos after every step])."""
    betas, alphas = beta_schedule(cfg)
    sig = sigmas_from_alphas(alphas)
    T = betas.numel()
    G = len(num_nodes_per_graph)
    if noise_from_time_t is not None:  # sampler.py:149-166
        seq = list(range(denoise_from_time_t - n_steps, denoise_from_time_t))
        alpha_t = alphas[denoise_from_time_t - 1]
        alpha_s = alphas[noise_from_time_t - 1] if noise_from_time_t != 0 else 1
        pos = pos_init + init_noise * ((1.0 - (alpha_t / alpha_s)) / alpha_t).sqrt()
    elif denoise_from_time_t is not None:  # sampler.py:168-177
        seq = list(range(denoise_from_time_t - n_steps, denoise_from_time_t))
        pos = pos_init
    else:  # sampler.py:179-182
        seq = list(range(T - n_steps, T))
        pos = pos_init * sig[-1]
    seq_next = [-1] + seq[:-1]
    traj = []
    for k, (i, j) in enumerate(zip(reversed(seq), reversed(seq_next))):
        edge_inv, ei, d = ensemble_forward(sds, cfg, atom_type, r_feat, p_feat, pos, bond_index,
                                           bond_type, num_nodes_per_graph)
        eps_pos = clip_norm(eq_transform(edge_inv, pos, ei, d), clip)
        noise = noises[k]
        if sampling_type == "ld":  # sampler.py:238-244
            step_size = step_lr * (sig[i] / 0.01) ** 2
            pos = pos + step_size * eps_pos / sig[i] + noise * torch.sqrt(step_size * 2)
        elif sampling_type == "ddpm":  # sampler.py:215-236
            t = torch.tensor([i])
            at = _compute_alpha(betas, t)
            atm1 = _compute_alpha(betas, torch.tensor([j]))
            beta_t = 1 - at / atm1
            e = -eps_pos
            pos_C = at.sqrt() * pos
            pos0 = (1.0 / at).sqrt() * pos_C - (1.0 / at - 1).sqrt() * e
            mean = ((atm1.sqrt() * beta_t) * pos0 + ((1 - beta_t).sqrt() * (1 - atm1)) * pos_C) / (1.0 - at)
            mask = 1 - float(i == 0)
            pos = (mean + mask * torch.exp(0.5 * beta_t.log()) * noise) / atm1.sqrt()
        else:
            raise NotImplementedError(sampling_type)
        if torch.isnan(pos).any():
            raise FloatingPointError()
        pos = center_pos(pos, batch, G)
        if clip_pos is not None:
            pos = torch.clamp(pos, min=-clip_pos, max=clip_pos)
        traj.append(pos.clone())
    return pos, traj


# ----------------------------------------------------------------------------
# training loss                                  reference condensenc.py:267-328
# ----------------------------------------------------------------------------
def get_loss(sd, cfg, atom_type, r_feat, p_feat, pos, bond_index, bond_type, batch,
             num_nodes_per_graph, time_step, pos_noise):
    """`time_step` (G,) and `pos_noise` (N,3) are the captured random draws."""
    _, alphas = beta_schedule(cfg)
    a = alphas.to(pos.dtype).index_select(0, time_step)
    a_pos = a.index_select(0, batch).unsqueeze(-1)
    pos_p = pos + pos_noise * (1.0 - a_pos).sqrt() / a_pos.sqrt()
    edge_inv, ei, d = forward(sd, cfg, atom_type, r_feat, p_feat, pos_p, bond_index, bond_type,
                              num_nodes_per_graph)
    node_eq = eq_transform(edge_inv, pos_p, ei, d)
    a_edge = a.index_select(0, batch.index_select(0, ei[0])).unsqueeze(-1)
    d_gt = get_distance(pos, ei).unsqueeze(-1)
    d_target = (d_gt - d) / (1.0 - a_edge).sqrt() * a_edge.sqrt()
    pos_target = eq_transform(d_target, pos_p, ei, d)
    return ((node_eq - pos_target) ** 2).sum(dim=-1, keepdim=True)


def to_torch_state(sd_np, dtype=torch.float32):
    return {k: torch.from_numpy(np.asarray(v)).to(dtype) for k, v in sd_np.items()}
