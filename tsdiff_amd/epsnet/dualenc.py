"""Drop-in for the reference `models/epsnet/dualenc.py::DualEncoderEpsNetwork` (GeoDiff legacy
network; `configs/geodiff_legacy/*.yml`).

Same constructor config, same `state_dict` keys (including the aliased `model_global.*` /
`model_local.*` entries, dualenc.py:168-204), same `forward` 6-tuple, `get_loss` and
`langevin_dynamics_sample` signatures.  Every computation runs in libtsdiff_hip.so: the
extended graph comes from the same topology / geometry kernels as the condensed network (one bond
graph instead of two), the global SchNet head and the local GINE head are chains of the C-ABI
training primitives (tsdiff_amd/train_ops.py), differentiable w.r.t. the parameters when autograd
is enabled.  Op-by-op (one launch per operation): this network is not reachable from the shipped
train.py / sampling.py, so it gets the functional form, not the fused kernels.

Not built (raise NotImplementedError): `type: dsm`, `is_sidechain` (the reference branch reads an
undefined `pos_gt`, dualenc.py:790-791), caller-supplied edge lists, `edge_encoder: gaussian` (the
reference class cannot be constructed, edge.py:28).
"""
import ctypes as C

import torch
from torch import nn

from .. import _lib, engine
from .._lib import check, ptr, stream_ptr
from ..encoder.gin import GINEncoder
from .condensenc import NUM_BOND_TYPES, _MLP, _MLPEdgeEncoder, _SchNetEncoder, _Swish, get_beta_schedule

_ACT_KIND = {"swish": 0, "ReLU": 2, "Softplus": 3}  # tsd_act_fwd kinds
_GINE_ACT = {None: 0, "ReLU": 1, "Softplus": 2}      # tsd_gine_csr_* activations


class _SchNetEncoderEmb(_SchNetEncoder):  # reference schnet.py:131-171 with embedding=True
    def __init__(self, hidden, num_convs):
        super().__init__(hidden, num_convs)
        self.node_emb = nn.Embedding(100, hidden, max_norm=10.0)


def _dual_cfg(config, extend_order, extend_radius):
    H = int(engine.cfg_get(config, "hidden_dim"))
    order = int(engine.cfg_get(config, "edge_order")) if extend_order else 1
    cutoff = float(engine.cfg_get(config, "cutoff"))
    return _lib.ModelCfg(hidden=H, num_convs=int(engine.cfg_get(config, "num_convs")), feat_dim=1,
                         edge_order=order, pred_edge_order=order,
                         edge_cutoff=cutoff if extend_radius else 0.0, conv_cutoff=cutoff,
                         smooth_conv=int(bool(engine.cfg_get(config, "smooth_conv", False))))


class DualEncoderEpsNetwork(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        if engine.cfg_get(config, "edge_encoder", "mlp") != "mlp":
            raise NotImplementedError("Unknown/unsupported edge encoder: %s" % engine.cfg_get(config, "edge_encoder"))
        self.mlp_act = engine.cfg_get(config, "mlp_act")
        if self.mlp_act not in _ACT_KIND:
            raise NotImplementedError(f"mlp_act={self.mlp_act}")
        H = int(engine.cfg_get(config, "hidden_dim"))
        self.edge_encoder_global = _MLPEdgeEncoder(H)
        self.edge_encoder_local = _MLPEdgeEncoder(H)
        self.encoder_global = _SchNetEncoderEmb(H, int(engine.cfg_get(config, "num_convs")))
        self.encoder_local = GINEncoder(hidden_dim=H, num_convs=int(engine.cfg_get(config, "num_convs_local")),
                                        embedding=True)
        self.grad_global_dist_mlp = _MLP(2 * H, [H, H // 2, 1])
        self.grad_local_dist_mlp = _MLP(2 * H, [H, H // 2, 1])

        self.model_type = engine.cfg_get(config, "type")
        if self.model_type != "diffusion":
            raise NotImplementedError(f"model type {self.model_type}: only `diffusion` is built")
        betas = get_beta_schedule(
            beta_schedule=engine.cfg_get(config, "beta_schedule"), beta_start=engine.cfg_get(config, "beta_start"),
            beta_end=engine.cfg_get(config, "beta_end"),
            num_diffusion_timesteps=engine.cfg_get(config, "num_diffusion_timesteps"))
        betas = torch.from_numpy(betas).float()
        self.betas = nn.Parameter(betas, requires_grad=False)
        self.alphas = nn.Parameter((1.0 - betas).cumprod(dim=0), requires_grad=False)
        self.num_timesteps = self.betas.size(0)
        self.TS = bool(engine.cfg_get(config, "TS", False))
        self.num_bond_types = NUM_BOND_TYPES

        global_modules = [self.edge_encoder_global, self.encoder_global, self.grad_global_dist_mlp]
        local_modules = [self.edge_encoder_local, self.encoder_local, self.grad_local_dist_mlp]
        if self.TS:
            self.edge_cat_act = engine.cfg_get(config, "edge_cat_act")
            if self.edge_cat_act not in _ACT_KIND:
                raise NotImplementedError(f"edge_cat_act={self.edge_cat_act}")
            self.edge_cat_global = nn.Sequential(nn.Linear(2 * H, H), _Swish(), nn.Linear(H, H))
            self.edge_cat_local = nn.Sequential(nn.Linear(2 * H, H), _Swish(), nn.Linear(H, H))
            global_modules.append(self.edge_cat_global)
            local_modules.append(self.edge_cat_local)
        self.model_global = nn.ModuleList(global_modules)
        self.model_local = nn.ModuleList(local_modules)
        self._batches = []

    # ------------------------------------------------------------------------------------------
    def device_batch(self, atom_type, bond_index, bond_type, batch, extend_order=True, extend_radius=True):
        ts = (atom_type, bond_index, bond_type, batch)  # (the cache entry keeps them alive: no address recycling)
        key = tuple((t._cdata, t._version) for t in ts) + (bool(extend_order), bool(extend_radius))
        for k, _, db in self._batches:
            if k == key:
                return db
        if atom_type.device.type != "cuda":
            raise _lib.TsdError("tsdiff_amd runs on the GPU only (inputs must be cuda tensors); there is no CPU fallback")
        cfg = _dual_cfg(self.config, extend_order, extend_radius)
        nb = self.num_bond_types
        bt = bond_type.to(torch.int64)
        # one bond graph: the k-hop orders are those of the union graph (common.py:255-325); feeding
        # it as reactant == product graph makes type_r the reference's embedding index (t | 22+k-1 | 0)
        union = torch.ones_like(bt) * (nb + 1) if self.TS else bt * nb + bt
        feat = torch.zeros(atom_type.shape[0], 1, dtype=torch.int64, device=atom_type.device)
        db = engine.DeviceBatch(cfg, atom_type, feat, feat, bond_index, union, batch)
        db.renorm_scratch = torch.zeros(100, dtype=torch.int32, device=atom_type.device)
        db.pair_code_raw = None
        if self.TS:  # the bonded pairs' own (r, p) types: the 5-bit fields of a second topology's pair codes
            lib = _lib.load()
            db.pair_code_raw = torch.zeros_like(db.pair_code)
            ng, pp = torch.zeros_like(db.node_graph), torch.zeros_like(db.pair_ptr)
            st = torch.zeros(4, dtype=torch.int32, device=atom_type.device)
            bi = bond_index.to(torch.int64).contiguous()
            check(lib.tsd_topology_build(db.N, db.G, db.P, int(bt.shape[0]), ptr(db.graph_ptr), ptr(db.pair_base),
                                         ptr(bi), ptr(bt.contiguous()), 1, db.max_n, ptr(ng), ptr(pp),
                                         ptr(db.pair_code_raw), ptr(st), stream_ptr()))
            if int(st[0].item()) & (_lib.STATUS_BAD_BOND | _lib.STATUS_ASYMMETRIC):
                raise ValueError("bond_index/bond_type: malformed bond list")
        self._batches = [(key, ts, db)] + self._batches[:2]
        return db

    def _heads(self, db, pos):
        from .. import train_ops as T
        return T.dual_forward(self, db, pos)

    def forward(self, atom_type, pos, bond_index, bond_type, batch, time_step, edge_index=None, edge_type=None,
                edge_length=None, return_edges=False, extend_order=True, extend_radius=True, is_sidechain=None):
        """reference dualenc.py:206-374"""
        if is_sidechain is not None:
            raise NotImplementedError("is_sidechain is not built")
        if edge_index is not None or edge_type is not None or edge_length is not None:
            raise NotImplementedError("caller-supplied edge lists are not built; the extended graph is derived "
                                      "from bond_index/pos as the reference does by default")
        db = self.device_batch(atom_type, bond_index, bond_type, batch, extend_order, extend_radius)
        pos_c = pos.detach().to(torch.float32).contiguous()
        out = self._heads(db, pos_c)
        E = db.enc.num_edges()
        um = db.enc.umap[:E].long()
        emb_type = out["emb_type_dir"]
        local_edge_mask = emb_type > 0  # dualenc.py:1222-1223
        edge_inv_global = out["s_global_u"].index_select(0, um).unsqueeze(-1)
        edge_inv_local = out["s_local_u"].index_select(0, um)[local_edge_mask].unsqueeze(-1)
        if not return_edges:
            return edge_inv_global, edge_inv_local
        edge_index, edge_length, _, _ = db.edges_to_torch("enc")
        return edge_inv_global, edge_inv_local, edge_index, out["edge_type_dir"], edge_length, local_edge_mask

    # ------------------------------------------------------------------------------------------
    def get_loss(self, atom_type, pos, bond_index, bond_type, batch, num_nodes_per_graph, num_graphs,
                 anneal_power=2.0, return_unreduced_loss=False, return_unreduced_edge_loss=False,
                 extend_order=True, extend_radius=True, is_sidechain=None, _time_step=None, _pos_noise=None):
        """reference dualenc.py:376-562 (`get_loss_diffusion`).  `_time_step` / `_pos_noise` inject the
        random draws (parity tests)."""
        from .. import train_ops as T
        if is_sidechain is not None:
            raise NotImplementedError("is_sidechain is not built")
        dev = pos.device
        node2graph = batch
        if _time_step is None:
            time_step = torch.randint(0, self.num_timesteps, size=(num_graphs // 2 + 1,), device=dev)
            time_step = torch.cat([time_step, self.num_timesteps - time_step - 1], dim=0)[:num_graphs]
        else:
            time_step = _time_step
        a = self.alphas.detach().index_select(0, time_step)
        a_pos = a.index_select(0, node2graph).unsqueeze(-1)
        pos_noise = torch.randn(size=pos.size(), device=dev) if _pos_noise is None else _pos_noise
        pos_perturbed = (pos + pos_noise * (1.0 - a_pos).sqrt() / a_pos.sqrt()).contiguous()
        db = self.device_batch(atom_type, bond_index, bond_type, batch, extend_order, extend_radius)
        out = self._heads(db, pos_perturbed)
        Eu = out["Eu"]
        src_u = db.enc_u.src[:Eu].long()
        a_edge = a.index_select(0, node2graph.index_select(0, src_u))
        d_perturbed = db.enc_u.dist[:Eu].clone()
        d_gt = T.pair_distance(db, "enc_u", pos, Eu)
        d_target = (d_gt - d_perturbed) / (1.0 - a_edge).sqrt() * a_edge.sqrt()
        local_u = out["emb_type_u"] > 0
        cutoff = float(engine.cfg_get(self.config, "cutoff"))
        global_mask = ((d_perturbed <= cutoff) | local_u) & ~local_u  # dualenc.py:504-510
        zero = torch.zeros_like(d_target)
        target_pos_global = T.EqUndirected.apply(torch.where(global_mask, d_target, zero), pos_perturbed, db)
        node_eq_global = T.EqUndirected.apply(torch.where(global_mask, out["s_global_u"], zero), pos_perturbed, db)
        loss_global = torch.sum((node_eq_global - target_pos_global) ** 2, dim=-1, keepdim=True)
        target_pos_local = T.EqUndirected.apply(torch.where(local_u, d_target, zero), pos_perturbed, db)
        node_eq_local = T.EqUndirected.apply(torch.where(local_u, out["s_local_u"], zero), pos_perturbed, db)
        loss_local = torch.sum((node_eq_local - target_pos_local) ** 2, dim=-1, keepdim=True)
        aa, bb = 2, 5
        loss = (aa * loss_global + bb * loss_local) / (aa + bb)
        if return_unreduced_edge_loss:
            return None  # the reference falls through (`pass`) and returns None as well (dualenc.py:556-557)
        if return_unreduced_loss:
            return loss, loss_global, loss_local
        return loss

    # ------------------------------------------------------------------------------------------
    def step_coefficients(self, n_steps, sampling_type, step_lr, eta=1.0):
        """(n_steps, 8) fp32 rows for tsd_sampler_step, one per iteration in execution order, evaluated with
        the reference's own fp32 tensor expressions (dualenc.py:776-949).  Returns (kinds, table, sigmas)."""
        alphas, betas = self.alphas.detach().cpu(), self.betas.detach().cpu()
        sigmas = (1.0 - alphas).sqrt() / alphas.sqrt()
        acp = (1 - torch.cat([torch.zeros(1), betas], dim=0)).cumprod(dim=0)
        seq = list(range(self.num_timesteps - n_steps, self.num_timesteps))
        seq_next = [-1] + seq[:-1]
        one, zero = torch.ones(()), torch.zeros(())
        rows, kinds = [], []
        for i, j in zip(reversed(seq), reversed(seq_next)):
            at, at_next = acp[i + 1], acp[j + 1]
            if sampling_type == "generalized":
                c1 = eta * ((1 - at / at_next) * (1 - at_next) / (1 - at)).sqrt()
                c2 = ((1 - at_next) - c1 ** 2).sqrt()
                step_ld = step_lr * (sigmas[i] / 0.01) ** 2 / sigmas[i]
                step_gen = 5 * ((1 - at).sqrt() / at.sqrt() - c2 / at_next.sqrt())
                step_pos = step_ld if step_ld < step_gen else step_gen
                noise_ld = torch.sqrt((step_lr * (sigmas[i] / 0.01) ** 2) * 2)
                noise_gen = 3 * (c1 / at_next.sqrt())
                step_noise = noise_ld if noise_ld < noise_gen else noise_gen
                rows.append(torch.stack([step_pos, one, step_noise] + [zero] * 5))  # pos + eps*step + noise*step_noise
                kinds.append(0)
            elif sampling_type in ("ddpm_noisy", "ddpm_det"):
                atm1 = at_next
                beta_t = 1 - at / atm1
                mask = 1.0 - float(i == 0)
                logvar = (beta_t * (1 - atm1) / (1 - at)).log() if sampling_type == "ddpm_det" else beta_t.log()
                rows.append(torch.stack([one, (1.0 / at).sqrt(), (1.0 / at - 1).sqrt(), atm1.sqrt() * beta_t,
                                         (1 - beta_t).sqrt() * (1 - atm1), 1.0 - at, mask * torch.exp(0.5 * logvar),
                                         one]))
                kinds.append(1)
            elif sampling_type == "ld":
                step_size = step_lr * (sigmas[i] / 0.01) ** 2
                rows.append(torch.stack([step_size, sigmas[i], torch.sqrt(step_size * 2)] + [zero] * 5))
                kinds.append(0)
            else:
                raise NotImplementedError(sampling_type)
        return kinds, torch.stack(rows).to(torch.float32).contiguous(), sigmas, seq

    def langevin_dynamics_sample(self, atom_type, pos_init, bond_index, bond_type, batch, num_graphs, extend_order,
                                 extend_radius=True, n_steps=100, step_lr=0.0000010, clip=1000, clip_local=None,
                                 clip_pos=None, min_sigma=0, is_sidechain=None, global_start_sigma=float("inf"),
                                 w_global=0.2, w_reg=1.0, **kwargs):
        """reference dualenc.py:687-967 (`langevin_dynamics_sample_diffusion`).  Extra keyword-only knobs:
        noises=(n_steps,N,3) injects the Gaussian draws, return_traj=False skips the trajectory."""
        from .. import train_ops as T
        lib = _lib.load()
        if is_sidechain is not None:
            raise NotImplementedError("is_sidechain is not built")
        sampling_type = kwargs.get("sampling_type", "ddpm_noisy")
        noises = kwargs.get("noises", None)
        return_traj = kwargs.get("return_traj", True)
        dev = pos_init.device
        kinds, coefs, sigmas, seq = self.step_coefficients(n_steps, sampling_type, step_lr, kwargs.get("eta", 1.0))
        coefs = coefs.to(dev)
        with torch.no_grad():
            pos = (pos_init * sigmas[-1].to(dev)).to(torch.float32).contiguous().clone()
            N = pos.shape[0]
            db = self.device_batch(atom_type, bond_index, bond_type, batch, extend_order, extend_radius)
            if int(num_graphs) != db.G:
                raise ValueError(f"num_graphs={num_graphs} but batch holds {db.G} graphs")
            if noises is None:
                noises = torch.randn(n_steps, N, 3, dtype=torch.float32, device=dev)
            noises = noises.to(device=dev, dtype=torch.float32).contiguous()
            db.status.zero_()
            eps_pos = torch.empty(N, 3, dtype=torch.float32, device=dev)
            eq_l = torch.empty(N, 3, dtype=torch.float32, device=dev)
            eq_g = torch.empty(N, 3, dtype=torch.float32, device=dev)
            traj = []
            for k, i in enumerate(reversed(seq)):
                out = self._heads(db, pos)
                local_u = out["emb_type_u"] > 0
                zero = torch.zeros_like(out["s_local_u"])
                s_l = torch.where(local_u, out["s_local_u"], zero).contiguous()
                check(lib.tsd_eq_und_fwd(N, db.out.struct(), ptr(pos), ptr(s_l), ptr(eq_l), stream_ptr()))
                use_global = bool(sigmas[i] < global_start_sigma)
                if use_global:  # edge_inv_global * (1 - local_edge_mask), dualenc.py:834-841
                    s_g = torch.where(local_u, zero, out["s_global_u"]).contiguous()
                    check(lib.tsd_eq_und_fwd(N, db.out.struct(), ptr(pos), ptr(s_g), ptr(eq_g), stream_ptr()))
                check(lib.tsd_dual_score(N, ptr(eq_l), ptr(eq_g) if use_global else None,
                                         float(-1.0 if clip_local is None else clip_local), float(clip),
                                         float(w_global), ptr(eps_pos), stream_ptr()))
                check(lib.tsd_sampler_step(kinds[k], N, db.G, ptr(db.graph_ptr), ptr(eps_pos), ptr(noises[k]),
                                           ptr(coefs[k]), float("inf"), float(-1.0 if clip_pos is None else clip_pos),
                                           ptr(pos), ptr(db.status), stream_ptr()))
                if return_traj:
                    traj.append(pos.clone())
            if int(db.status[0].item()) & _lib.STATUS_NAN:  # sticky device flag, read once
                print("NaN detected. Please restart.")
                raise FloatingPointError()
            pos_traj = [p.cpu() for p in traj]
        return pos, pos_traj

    langevin_dynamics_sample_diffusion = langevin_dynamics_sample


def is_local_edge(edge_type):  # reference dualenc.py:1222-1223
    return edge_type > 0
