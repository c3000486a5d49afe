"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the GeoDiff legacy dual-encoder network
(reference models/epsnet/dualenc.py).  Same rules as oracle/tsdiff_oracle.py:
only tests/, smoke() and bench.py's cpu_baseline may import it; it is a restatement (our own code,
torch-CPU, no PyG) and it is PINNED against goldens that oracle/gen_golden.py produced by running
the unchanged reference `DualEncoderEpsNetwork` (tests/test_oracle_golden.py).
"""
import numpy as np
import torch
import torch.nn.functional as F

from .tsdiff_oracle import (NUM_BOND_TYPES, _hop_matrix, beta_schedule, center_pos, clip_norm, eq_transform,
                            get_distance, schnet_encoder, sigmas_from_alphas, swish)

_ACT = {"ReLU": F.relu, "swish": swish, "Softplus": F.softplus}


def extend_graph(pos, bond_index, bond_type, num_nodes_per_graph, order, cutoff, extend_order=True,
                 extend_radius=True):
    """reference extend_graph_order_radius (common.py:387-417) = _extend_graph_order (:255-325; types:
    bond t, 22^2 + k - 1 for a k-hop pair) then _extend_to_radius_graph (:328-384; radius-only pairs get 0).
    Returns edge_index (2,E) row-major sorted, edge_type (E,)."""
    pos_np = np.asarray(pos, dtype=np.float32)
    bi, bt = np.asarray(bond_index), np.asarray(bond_type)
    off = np.concatenate([[0], np.cumsum(num_nodes_per_graph)])
    c2 = np.float32(cutoff) * np.float32(cutoff)
    rows, cols, types = [], [], []
    for g, n in enumerate(num_nodes_per_graph):
        lo, hi = off[g], off[g + 1]
        sel = (bi[0] >= lo) & (bi[0] < hi)
        s, d, t = bi[0, sel] - lo, bi[1, sel] - lo, bt[sel]
        tm = np.zeros((n, n), dtype=np.int64)
        np.add.at(tm, (s, d), t)  # to_dense_adj sums duplicates
        if extend_order:
            hop = _hop_matrix(n, s, d, order)
            high = np.where(hop > 1, NUM_BOND_TYPES ** 2 + hop - 1, 0)
            assert (tm * high == 0).all()  # common.py:306
            tm = tm + high
        member = tm != 0
        if extend_radius:
            p = pos_np[lo:hi]
            diff = p[:, None, :] - p[None, :, :]
            member = member | ((diff * diff).sum(-1, dtype=np.float32) < c2)
        member &= ~np.eye(n, dtype=bool)
        r, c = np.nonzero(member)
        rows.append(r + lo)
        cols.append(c + lo)
        types.append(tm[r, c])
    ei = torch.from_numpy(np.stack([np.concatenate(rows), np.concatenate(cols)]).astype(np.int64))
    return ei, torch.from_numpy(np.concatenate(types).astype(np.int64))


def _mlp(sd, prefix, x, act, n):  # common.py:74-90
    for i in range(n):
        x = F.linear(x, sd[f"{prefix}.layers.{i}.weight"], sd[f"{prefix}.layers.{i}.bias"])
        if i < n - 1:
            x = act(x)
    return x


def _edge_attr(sd, cfg, which, d, edge_type):
    """dualenc.py:262-283 (global) / 305-326 (local) + edge.py:58-68"""
    nb = NUM_BOND_TYPES
    act = _ACT[cfg["mlp_act"]]
    enc = f"edge_encoder_{which}"
    d_emb = _mlp(sd, enc + ".mlp", d, act, 2)
    emb = sd[enc + ".bond_emb.weight"]
    mask = edge_type // nb ** 2 == 0
    high = torch.where(~mask, edge_type % nb ** 2 + nb, torch.zeros_like(edge_type))
    if cfg.get("TS", False):
        t1 = torch.where(mask, edge_type // nb, torch.zeros_like(edge_type)) + high
        t2 = torch.where(mask, edge_type % nb, torch.zeros_like(edge_type)) + high
        cat = torch.cat([d_emb * emb[t1], d_emb * emb[t2]], dim=-1)
        c = f"edge_cat_{which}"
        x = _ACT[cfg["edge_cat_act"]](F.linear(cat, sd[c + ".0.weight"], sd[c + ".0.bias"]))
        return F.linear(x, sd[c + ".2.weight"], sd[c + ".2.bias"])
    t1 = torch.where(mask, edge_type % nb, torch.zeros_like(edge_type)) + high
    return d_emb * emb[t1]


def embedding_max_norm(weight, idx, max_norm=10.0):
    """torch.nn.Embedding(max_norm): looked-up rows with norm > max_norm are rescaled (in the reference: in
    place, schnet.py:151); returns the rescaled table."""
    w = weight.clone()
    with torch.no_grad():
        rows = torch.unique(idx)
        norm = w[rows].norm(dim=1)
        scale = torch.where(norm > max_norm, max_norm / (norm + 1e-7), torch.ones_like(norm))
        w[rows] = w[rows] * scale.unsqueeze(-1)
    return w


def gin_encoder(sd, prefix, z, edge_index, edge_attr, num_convs, act=F.relu):
    """reference gin.py:79-149 with GINEConv (:19-76), eps buffer = 0, short_cut, embedding"""
    x = sd[prefix + "node_emb.weight"][z]
    for k in range(num_convs):
        msg = act(x[edge_index[0]] + edge_attr)
        out = torch.zeros_like(x).index_add_(0, edge_index[1], msg)
        out = out + (1 + sd[f"{prefix}convs.{k}.eps"]) * x
        q = f"{prefix}convs.{k}.nn"
        h = F.linear(act(F.linear(out, sd[q + ".layers.0.weight"], sd[q + ".layers.0.bias"])),
                     sd[q + ".layers.1.weight"], sd[q + ".layers.1.bias"])
        if k < num_convs - 1:
            h = act(h)
        x = h + x
    return x


def forward(sd, cfg, atom_type, pos, bond_index, bond_type, num_nodes_per_graph, extend_order=True,
            extend_radius=True):
    """reference DualEncoderEpsNetwork.forward (dualenc.py:206-374), `diffusion` type.
    Returns the 6-tuple (edge_inv_global, edge_inv_local, edge_index, edge_type, edge_length, local_mask)."""
    act = _ACT[cfg["mlp_act"]]
    ei, et = extend_graph(pos, bond_index, bond_type, num_nodes_per_graph, int(cfg["edge_order"]),
                          float(cfg["cutoff"]), extend_order, extend_radius)
    d = get_distance(pos, ei).unsqueeze(-1)
    local = et > 0
    ea_g = _edge_attr(sd, cfg, "global", d, et)
    w = embedding_max_norm(sd["encoder_global.node_emb.weight"], atom_type)
    h = schnet_encoder(sd, w[atom_type], ei, d.view(-1), ea_g, int(cfg["num_convs"]), float(cfg["cutoff"]),
                       smooth=bool(cfg.get("smooth_conv", False)), prefix="encoder_global.")
    inv_g = _mlp(sd, "grad_global_dist_mlp", torch.cat([h[ei[0]] * h[ei[1]], ea_g], dim=-1), act, 3)
    ea_l = _edge_attr(sd, cfg, "local", d, et)
    eil, eal = ei[:, local], ea_l[local]
    hl = gin_encoder(sd, "encoder_local.", atom_type, eil, eal, int(cfg["num_convs_local"]))
    inv_l = _mlp(sd, "grad_local_dist_mlp", torch.cat([hl[eil[0]] * hl[eil[1]], eal], dim=-1), act, 3)
    return inv_g, inv_l, ei, et, d, local


def get_loss(sd, cfg, atom_type, pos, bond_index, bond_type, batch, num_nodes_per_graph, time_step, pos_noise,
             extend_order=True, extend_radius=True):
    """reference get_loss_diffusion (dualenc.py:425-562) with the captured random draws.
    Returns (loss, loss_global, loss_local), each (N,1)."""
    _, alphas = beta_schedule(cfg)
    a = alphas.index_select(0, time_step)
    a_pos = a.index_select(0, batch).unsqueeze(-1)
    pos_p = pos + pos_noise * (1.0 - a_pos).sqrt() / a_pos.sqrt()
    inv_g, inv_l, ei, et, d, local = forward(sd, cfg, atom_type, pos_p, bond_index, bond_type, num_nodes_per_graph,
                                             extend_order, extend_radius)
    a_edge = a.index_select(0, batch.index_select(0, ei[0])).unsqueeze(-1)
    d_gt = get_distance(pos, ei).unsqueeze(-1)
    d_target = (d_gt - d) / (1.0 - a_edge).sqrt() * a_edge.sqrt()
    lm = local.unsqueeze(-1)
    gmask = ((d <= float(cfg["cutoff"])) | lm) & ~lm
    zero = torch.zeros_like(d_target)
    tgt_g = eq_transform(torch.where(gmask, d_target, zero), pos_p, ei, d)
    eq_g = eq_transform(torch.where(gmask, inv_g, zero), pos_p, ei, d)
    loss_g = ((eq_g - tgt_g) ** 2).sum(dim=-1, keepdim=True)
    tgt_l = eq_transform(d_target[local], pos_p, ei[:, local], d[local])
    eq_l = eq_transform(inv_l, pos_p, ei[:, local], d[local])
    loss_l = ((eq_l - tgt_l) ** 2).sum(dim=-1, keepdim=True)
    return (2 * loss_g + 5 * loss_l) / 7, loss_g, loss_l


def sample(sd, cfg, atom_type, pos_init, bond_index, bond_type, batch, num_nodes_per_graph, noises, n_steps,
           step_lr=1e-6, clip=1000.0, clip_local=None, clip_pos=None, global_start_sigma=float("inf"), w_global=0.2,
           sampling_type="ddpm_noisy", eta=1.0, extend_order=True, extend_radius=True):
    """reference langevin_dynamics_sample_diffusion (dualenc.py:754-967) with the per-step noise injected."""
    betas, alphas = beta_schedule(cfg)
    sig = sigmas_from_alphas(alphas)
    acp = (1 - torch.cat([torch.zeros(1), betas], dim=0)).cumprod(dim=0)
    T, G = betas.numel(), len(num_nodes_per_graph)
    seq = list(range(T - n_steps, T))
    seq_next = [-1] + seq[:-1]
    pos = pos_init * sig[-1]
    traj = []
    for k, (i, j) in enumerate(zip(reversed(seq), reversed(seq_next))):
        inv_g, inv_l, ei, et, d, local = forward(sd, cfg, atom_type, pos, bond_index, bond_type, num_nodes_per_graph,
                                                 extend_order, extend_radius)
        eq_l = eq_transform(inv_l, pos, ei[:, local], d[local])
        if clip_local is not None:
            eq_l = clip_norm(eq_l, clip_local)
        if sig[i] < global_start_sigma:
            eq_g = clip_norm(eq_transform(inv_g * (1 - local.view(-1, 1).float()), pos, ei, d), clip)
        else:
            eq_g = 0
        eps_pos = eq_l + eq_g * w_global
        noise = noises[k]
        at, at_next = acp[i + 1], acp[j + 1]
        if sampling_type == "generalized":
            et_ = -eps_pos
            c1 = eta * ((1 - at / at_next) * (1 - at_next) / (1 - at)).sqrt()
            c2 = ((1 - at_next) - c1 ** 2).sqrt()
            s_ld = step_lr * (sig[i] / 0.01) ** 2 / sig[i]
            s_gen = 5 * ((1 - at).sqrt() / at.sqrt() - c2 / at_next.sqrt())
            step_pos = s_ld if s_ld < s_gen else s_gen
            n_ld = torch.sqrt((step_lr * (sig[i] / 0.01) ** 2) * 2)
            n_gen = 3 * (c1 / at_next.sqrt())
            step_noise = n_ld if n_ld < n_gen else n_gen
            pos = pos - et_ * step_pos + noise * step_noise
        elif sampling_type in ("ddpm_noisy", "ddpm_det"):
            atm1 = at_next
            beta_t = 1 - at / atm1
            e = -eps_pos
            pos0 = (1.0 / at).sqrt() * pos - (1.0 / at - 1).sqrt() * e
            mean = ((atm1.sqrt() * beta_t) * pos0 + ((1 - beta_t).sqrt() * (1 - atm1)) * pos) / (1.0 - at)
            mask = 1 - float(i == 0)
            logvar = (beta_t * (1 - atm1) / (1 - at)).log() if sampling_type == "ddpm_det" else beta_t.log()
            pos = mean + mask * torch.exp(0.5 * logvar) * noise
        elif sampling_type == "ld":
            step = step_lr * (sig[i] / 0.01) ** 2
            pos = pos + step * eps_pos / sig[i] + noise * torch.sqrt(step * 2)
        else:
            raise NotImplementedError(sampling_type)
        if torch.isnan(pos).any():
            raise FloatingPointError()
        pos = center_pos(pos, batch, G)
        if clip_pos is not None:
            pos = torch.clamp(pos, min=-clip_pos, max=clip_pos)
        traj.append(pos.clone())
    return pos, traj
