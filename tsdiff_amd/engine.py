"""Host-side plumbing between torch device tensors and the C ABI (include/tsdiff_hip.h).

torch is used for device memory (caching allocator), streams and dtype conversion only; every
computation of the path runs in libtsdiff_hip.so.
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

from . import _lib
from ._lib import Batch, Edges, Geometry, ModelCfg, check, ptr, stream_ptr

# order of the flattened reference state_dict expected by tsd_pack_weights (include/tsdiff_hip.h)


def raw_param_names(num_convs):
    names = [
        "edge_encoder.bond_emb.weight",
        "edge_encoder.mlp.layers.0.weight", "edge_encoder.mlp.layers.0.bias",
        "edge_encoder.mlp.layers.1.weight", "edge_encoder.mlp.layers.1.bias",
        "atom_embedding.weight", "atom_feat_embedding.weight",
    ]
    for l in range(num_convs):
        p = f"encoder.interactions.{l}."
        names += [p + "conv.lin1.weight", p + "conv.lin2.weight", p + "conv.lin2.bias",
                  p + "conv.nn.0.weight", p + "conv.nn.0.bias", p + "conv.nn.2.weight", p + "conv.nn.2.bias",
                  p + "lin.weight", p + "lin.bias"]
    names += [f"grad_dist_mlp.layers.{i}.{k}" for i in range(3) for k in ("weight", "bias")]
    names += ["edge_cat.0.weight", "edge_cat.0.bias", "edge_cat.2.weight", "edge_cat.2.bias"]
    return names


def cfg_get(cfg, key, default=None):
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


def make_cfg(model_config):
    """tsd_model_cfg from the reference's `config.model` (EasyDict / dict / attribute object)."""
    enc = cfg_get(model_config, "encoder")
    H = int(cfg_get(model_config, "hidden_dim"))
    if int(cfg_get(enc, "hidden_dim")) != H:
        raise ValueError("condensenc needs encoder.hidden_dim == hidden_dim")
    if cfg_get(enc, "name", "schnet") != "schnet":
        raise NotImplementedError("Unknown/unsupported encoder: %s" % cfg_get(enc, "name"))
    if cfg_get(model_config, "edge_encoder", "mlp") != "mlp":
        raise NotImplementedError("Unknown/unsupported edge encoder: %s" % cfg_get(model_config, "edge_encoder"))
    for k in ("mlp_act", "edge_cat_act"):
        if cfg_get(model_config, k, "swish") != "swish":
            raise NotImplementedError(f"{k}={cfg_get(model_config, k)}: only swish is built")
    return ModelCfg(
        hidden=H,
        num_convs=int(cfg_get(enc, "num_convs")),
        feat_dim=int(cfg_get(model_config, "feat_dim")),
        edge_order=int(cfg_get(model_config, "edge_order")),
        pred_edge_order=int(cfg_get(model_config, "pred_edge_order")),
        edge_cutoff=float(cfg_get(model_config, "edge_cutoff")),
        conv_cutoff=float(cfg_get(enc, "cutoff")),
        smooth_conv=int(bool(cfg_get(enc, "smooth_conv", False))),
    )


def cfg_tuple(cfg):
    """the fields of a tsd_model_cfg as a comparable tuple"""
    return tuple(getattr(cfg, f) for f, _ in cfg._fields_)


def pack_weights(cfg, named_tensors, device):
    """named_tensors: name -> tensor (reference state_dict names).  Returns packed fp32 device arena."""
    lib = _lib.load()
    names = raw_param_names(cfg.num_convs)
    raw = torch.cat([named_tensors[n].detach().to(device=device, dtype=torch.float32).reshape(-1) for n in names])
    n_raw = lib.tsd_raw_weight_floats(C.byref(cfg))
    if raw.numel() != n_raw:
        raise ValueError(f"state_dict has {raw.numel()} weights, config needs {n_raw}")
    packed = torch.zeros(lib.tsd_packed_weight_floats(C.byref(cfg)), dtype=torch.float32, device=device)
    check(lib.tsd_pack_weights(C.byref(cfg), ptr(raw), ptr(packed), stream_ptr()))
    return packed


_SIDE_STREAMS = {}
MAX_TYPE_BUCKETS = 96  # (type_r, type_p) pairs per batch whose folded matrices are kept (H = 256: 25 MB per checkpoint)
# the A/B and cross-check switches of the host side (arithmetic of the tile GEMMs, typed embedding tiles, one-launch
# forward, filter tile width, fused step tail, fused encoder) live in tsdiff_amd.options, documented there
from .options import OPTIONS  # noqa: E402


def _side_stream(device):
    key = str(device)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]


class _ZeroArena:
    """The buffers of a batch carved from ONE zero-initialised allocation (one fill launch instead of ~20) and one
    uninitialised one: a fresh batch per training step pays for every launch and allocation of its construction."""
    ALIGN = 256
    ITEM = {torch.int32: 4, torch.float32: 4, torch.int16: 2, torch.uint8: 1, torch.int64: 8}

    @staticmethod
    def need(n, dtype):
        return (n * _ZeroArena.ITEM[dtype] + _ZeroArena.ALIGN - 1) & ~(_ZeroArena.ALIGN - 1)

    def __init__(self, device, nbytes, zero=True):
        self.buf = (torch.zeros if zero else torch.empty)(max(nbytes, 1), dtype=torch.uint8, device=device)
        self.o = 0

    def take(self, n, dtype):
        sz = n * self.ITEM[dtype]
        o = self.o
        self.o = (o + sz + self.ALIGN - 1) & ~(self.ALIGN - 1)
        if self.o > self.buf.numel():
            raise RuntimeError("internal: zero arena undersized")
        return self.buf[o:o + sz].view(dtype)


class EdgeList:
    """device buffers of one extended-graph edge list (capacity = num_pairs)"""

    @staticmethod
    def zero_bytes(N):
        return _ZeroArena.need(1, torch.int32) + _ZeroArena.need(N + 1, torch.int32)

    @staticmethod
    def raw_bytes(P):
        cap = max(P, 1) + _lib.EDGE_PAD
        return 5 * _ZeroArena.need(cap, torch.int32) + 2 * _ZeroArena.need(cap, torch.uint8)

    def __init__(self, N, P, device, arena, raw):
        self.count = arena.take(1, torch.int32)
        self.row_ptr = arena.take(N + 1, torch.int32)
        # entries [0, count) of the per-edge arrays are rewritten by every geometry build and nothing uses values
        # past count: no fill launches for them (a fresh batch per training step builds 35 of these).  Every array
        # carries TSD_EDGE_PAD spare entries: the node role fetches dst / umap 8 edges at a time by scalar loads.
        cap = max(P, 1) + _lib.EDGE_PAD
        self.src = raw.take(cap, torch.int32)
        self.dst = raw.take(cap, torch.int32)
        self.dist = raw.take(cap, torch.float32)
        self.type_r = raw.take(cap, torch.uint8)
        self.type_p = raw.take(cap, torch.uint8)
        self.pair_id = raw.take(cap, torch.int32)
        self.umap = raw.take(cap, torch.int32)

    def struct(self):
        return Edges(*[C.c_void_p(t.data_ptr()) for t in (
            self.count, self.row_ptr, self.src, self.dst, self.dist, self.type_r, self.type_p, self.pair_id,
            self.umap)])

    def num_edges(self):
        return int(self.count.item())  # host sync


class DeviceBatch:
    """Everything pos-independent about one batch of reaction graphs, resident in HBM:
    topology (k-hop pair codes), edge-list buffers, per-checkpoint node embeddings, workspace.

    Built once per batch (the reference recomputes all of it 5000 x M times)."""

    def __init__(self, cfg, atom_type, r_feat, p_feat, bond_index, bond_type, batch=None,
                 num_nodes_per_graph=None, defer_status=False):
        lib = _lib.load()
        dev = atom_type.device
        if dev.type != "cuda":
            raise _lib.TsdError("tsdiff_amd runs on the GPU only (inputs must be cuda tensors); "
                                "there is no CPU fallback")
        self.cfg = cfg
        self.device = dev
        N = int(atom_type.shape[0])
        if num_nodes_per_graph is None:
            if batch is None:
                raise ValueError("need `batch` or `num_nodes_per_graph`")
            nn_host = torch.bincount(batch).cpu().numpy().astype(np.int64) if N else np.zeros(0, np.int64)
        else:
            nn_host = (num_nodes_per_graph.detach().cpu().numpy() if torch.is_tensor(num_nodes_per_graph)
                       else np.asarray(num_nodes_per_graph)).astype(np.int64)
        if int(nn_host.sum()) != N:
            raise ValueError("num_nodes_per_graph does not sum to the number of atoms")
        G = int(nn_host.shape[0])
        graph_ptr = np.zeros(G + 1, np.int64)
        np.cumsum(nn_host, out=graph_ptr[1:])
        pair_base = np.zeros(G + 1, np.int64)
        np.cumsum(nn_host * (nn_host - 1), out=pair_base[1:])
        P = int(pair_base[-1])
        if P * cfg.hidden >= 2 ** 31 or N * cfg.hidden >= 2 ** 31:
            # the library's own guard (api.hip capacity_ok), checked here before anything large is allocated
            raise NotImplementedError(f"batch too large: {P} ordered pairs x hidden {cfg.hidden} >= 2^31 elements per "
                                      "edge matrix (largest verified size: 1024 x 64-atom graphs); split the batch "
                                      "(tsdiff_amd.distributed shards graphs)")
        self.N, self.G, self.P = N, G, P
        self.max_n = int(nn_host.max()) if G else 0
        self.num_nodes_per_graph_host = nn_host
        self.graph_ptr = torch.from_numpy(graph_ptr.astype(np.int32)).to(dev)
        self.pair_base = torch.from_numpy(pair_base.astype(np.int32)).to(dev)
        i32, f32 = torch.int32, torch.float32
        zeros = [("node_graph", max(N, 1), i32), ("pair_ptr", N + 1, i32), ("pair_code", max(P, 1), torch.int16),
                 # tsd_sampler_state: [0] status flags (also the topology build's status word), [1] step counter, run args
                 ("status", _lib.SAMPLER_STATE_INTS, i32),
                 ("typed_counts", 4, i32),  # tsd_typed_tiles_build's four counts (read back with the status word)
                 ("attr_row", max(P // 2, 1), i32), ("pair2out", max(P, 1), i32), ("pair2u", max(2 * P, 1), i32),
                 ("geo_scratch", int(lib.tsd_geometry_scratch_ints(N, P)), i32),
                 ("pos_work", 3 * max(N, 1), f32),  # the loop's in-place positions
                 ("scratch", ((P + 63) // 64) * 64 + 3 * N + 128, f32)]
        arena = _ZeroArena(dev, sum(_ZeroArena.need(n, dt) for _, n, dt in zeros) + 5 * EdgeList.zero_bytes(N))
        for name, n, dt in zeros:
            setattr(self, name, arena.take(n, dt))
        self.pos_work = self.pos_work.view(max(N, 1), 3)
        self.atom_type = atom_type.to(torch.int64).contiguous()
        self.r_feat = r_feat.to(torch.int64).contiguous()
        self.p_feat = p_feat.to(torch.int64).contiguous()
        bond_index = bond_index.to(torch.int64).contiguous()
        bond_type = bond_type.to(torch.int64).contiguous()
        nb = int(bond_type.shape[0])
        max_order = max(cfg.edge_order, cfg.pred_edge_order)
        check(lib.tsd_topology_build(N, G, P, nb, ptr(self.graph_ptr), ptr(self.pair_base), ptr(bond_index),
                                     ptr(bond_type), max_order, self.max_n, ptr(self.node_graph),
                                     ptr(self.pair_ptr), ptr(self.pair_code), ptr(self.status), stream_ptr()))
        # static type-sorted embedding tiles (inference forward: one GEMM per embedded edge instead of three,
        # include/tsdiff_hip.h tsd_typed_tiles): built on the device behind the topology, their four counts come back
        # with the status word in the same host read
        self.typed = None
        if OPTIONS.typed_tiles and not defer_status and P > 0:
            cap = int(lib.tsd_typed_tiles_capacity(P))
            PU = P // 2
            tt = torch.empty(6 * PU + 6 * cap + 2 * 1024 + 8192, dtype=i32, device=dev)
            v = [tt[k * PU:(k + 1) * PU] for k in range(6)]
            o = 6 * PU
            enc_tile, diff_tile = tt[o:o + 3 * cap], tt[o + 3 * cap:o + 6 * cap]
            keys = tt[o + 6 * cap:o + 6 * cap + 2048]
            scratch = tt[o + 6 * cap + 2048:]
            counts = self.typed_counts
            check(lib.tsd_typed_tiles_build(C.byref(cfg), N, P, ptr(self.graph_ptr), ptr(self.node_graph),
                                            ptr(self.pair_ptr), ptr(self.pair_code), ptr(v[0]), ptr(v[1]), ptr(v[2]),
                                            ptr(enc_tile), ptr(v[3]), ptr(v[4]), ptr(v[5]), ptr(diff_tile), ptr(keys),
                                            ptr(counts), ptr(scratch), stream_ptr()))
            self.typed = dict(buf=tt, cap=cap, v=v, enc_tile=enc_tile, diff_tile=diff_tile, keys=keys)
        # the topology status word needs a host read: now (one sync per batch), or -- training -- together with
        # the edge counts that tsd_train_forward reads anyway
        self.status_pending = True
        if not defer_status:
            words = torch.cat([self.status[:1], self.typed_counts]).cpu()  # ONE host read per batch
            self.check_status(int(words[0]))
            if self.typed is not None:
                nt_e, nb_e, nt_d, nb_d = (int(x) for x in words[1:5])
                if nt_e == 0 or nb_e + nb_d > MAX_TYPE_BUCKETS:
                    self.typed = None  # (no pairs, or an unusually rich type mix: the generic embedding kernel)
                else:
                    t = self.typed
                    t.update(nt_e=nt_e, nb_e=nb_e, nt_d=nt_d, nb_d=nb_d,
                             slot_keys=torch.cat([t["keys"][:nb_e], t["keys"][1024:1024 + nb_d]]).contiguous())
        raw = _ZeroArena(dev, 2 * EdgeList.raw_bytes(P) + 3 * EdgeList.raw_bytes(P // 2), zero=False)
        self.enc = EdgeList(N, P, dev, arena, raw)      # directed lists: the reference's edge_index order
        self.out = EdgeList(N, P, dev, arena, raw)
        self.enc_u = EdgeList(N, P // 2, dev, arena, raw)  # undirected (src < dst) lists the per-edge MLPs run on
        self.out_u = EdgeList(N, P // 2, dev, arena, raw)
        self.diff_u = EdgeList(N, P // 2, dev, arena, raw)
        self.workspace = None
        self.edge_inv = None
        self.z = None
        self._z_key = None
        self.unit_node = None   # unit partition of the fused per-unit encoder (built at bind time: depends on M)
        self.units_single_graph = False
        self._units_for = None
        self.geo_gen = 0  # bumped whenever the edge lists are rebuilt (saved training contexts check it)
        self.ready_event = None  # set by a build on a side stream (prefetch): consumers wait for it, stream to stream
        self._plans = {}  # (kind, clip, clip_pos) -> tsd_sampler_plan* of the bound checkpoints
        self._plan_streams = {}  # streams the plans were launched on (drop_plans waits for them)
        self.gemm = None  # None: OPTIONS.gemm; "f32" after a range fallback on this batch
        self.per_block = False  # True after an in-launch wait of the one-launch forward gave up on this batch
        self.test_flags = 0     # extra tsd_batch.reserved bits (tests: bit 3 = fault injection in the one-launch forward)

    def owned_tensors(self):
        """every device tensor this object holds right now (arenas, inputs, typed-tile buffers): what a build on a
        side stream hands to the allocator's cross-stream bookkeeping (`Tensor.record_stream`)"""
        seen, out = set(), []

        def walk(o, depth=0):
            if torch.is_tensor(o):
                if o.is_cuda and o.untyped_storage().data_ptr() not in seen:
                    seen.add(o.untyped_storage().data_ptr())
                    out.append(o)
            elif isinstance(o, dict):
                for v in o.values():
                    walk(v, depth + 1)
            elif isinstance(o, (list, tuple)):
                for v in o:
                    walk(v, depth + 1)
            elif isinstance(o, EdgeList) and depth < 3:
                walk(vars(o), depth + 1)
        walk(vars(self))
        return out

    def check_status(self, word=None):
        if not self.status_pending:
            return
        st = int(self.status[0].item()) if word is None else int(word)
        self.status_pending = False
        if st & _lib.STATUS_BAD_BOND:
            raise ValueError("bond_index/bond_type: self loop, bond across graphs or value out of range")
        if st & _lib.STATUS_ASYMMETRIC:
            raise ValueError("bond list must contain both directions of every bond with equal types "
                             "(reference utils/datasets.py:491-507)")

    # ---- per-checkpoint state ----------------------------------------------------------------
    def bind_models(self, packed_list, key):
        """packed_list: list of M packed weight arenas (same config). Computes z per checkpoint."""
        lib = _lib.load()
        if self._z_key == key:
            return
        self.drop_plans()  # captured graphs hold the old weight / workspace pointers
        M = len(packed_list)
        H = self.cfg.hidden
        self.weights = torch.stack(packed_list) if M > 1 else packed_list[0].reshape(1, -1)
        self.z = torch.empty(M, max(self.N, 1), H, dtype=torch.float32, device=self.device)
        self.x1_0 = torch.empty(M, max(self.N, 1), H, dtype=torch.float32, device=self.device)
        for m in range(M):  # both are pos independent: once per batch and checkpoint
            check(lib.tsd_node_embed(C.byref(self.cfg), ptr(self.weights[m]), self.N, ptr(self.atom_type),
                                     ptr(self.r_feat), ptr(self.p_feat), ptr(self.z[m]), stream_ptr()))
            check(lib.tsd_node_lin1(C.byref(self.cfg), ptr(self.weights[m]), 0, self.N, ptr(self.z[m]),
                                    ptr(self.x1_0[m]), stream_ptr()))
        nws = lib.tsd_forward_workspace_floats(C.byref(self.cfg), self.N, self.P, M)
        if self.workspace is None or self.workspace.numel() < nws:
            self.workspace = torch.empty(max(nws, 1), dtype=torch.float32, device=self.device)
        self.edge_inv_u = torch.zeros(M, max(self.P // 2, 1), dtype=torch.float32, device=self.device)
        self.bucket_weights = None
        self.weights16 = None
        self.bucket_weights16 = None
        if self.typed is not None and OPTIONS.typed_tiles:
            t = self.typed
            ns = t["nb_e"] + t["nb_d"]
            per = int(lib.tsd_bucket_weights_floats(C.byref(self.cfg), ns))
            self.bucket_weights = torch.empty(M, per, dtype=torch.float32, device=self.device)
            for m in range(M):  # folded once per (batch, checkpoint), fp64 accumulation
                check(lib.tsd_bucket_weights_build(C.byref(self.cfg), ptr(self.weights[m]), ns, ptr(t["slot_keys"]),
                                                   ptr(self.bucket_weights[m]), stream_ptr()))
            # the f16-plane images of both arenas (split-f16 forward: needs the typed embedding)
            self.weights16 = torch.empty_like(self.weights)
            self.bucket_weights16 = torch.empty_like(self.bucket_weights)
            for m in range(M):
                check(lib.tsd_pack_weights16(C.byref(self.cfg), ptr(self.weights[m]), ptr(self.weights16[m]),
                                             stream_ptr()))
                check(lib.tsd_bucket_weights16(C.byref(self.cfg), ptr(self.bucket_weights[m]), ns,
                                               ptr(self.bucket_weights16[m]), stream_ptr()))
        self.M = M
        self._z_key = key
        self.build_units(M)

    def build_units(self, M):
        """the unit partition of the fused per-unit encoder for M bound checkpoints (host knowledge, pos independent)"""
        key = (M, str(OPTIONS.fused_encoder))
        if self._units_for == key:
            return
        self._units_for = key
        self.unit_node = None
        if self.G == 0 or self.max_n > _lib.UNIT_MAX_NODES or self.P == 0:
            return
        cus = torch.cuda.get_device_properties(self.device).multi_processor_count
        bounds = partition_units(self.num_nodes_per_graph_host, M, cus)
        self.unit_node = torch.from_numpy(bounds.astype(np.int32)).to(self.device)
        # every unit one graph (certain when every graph has more than half a unit's atoms): the block-tile forms apply
        self.units_single_graph = bool(len(bounds) - 1 == self.G and
                                       int(self.num_nodes_per_graph_host.min()) > _lib.UNIT_MAX_NODES // 2)

    def geo_struct(self):
        return Geometry(enc=self.enc.struct(), out=self.out.struct(), enc_u=self.enc_u.struct(),
                        out_u=self.out_u.struct(), diff_u=self.diff_u.struct(),
                        attr_row=self.attr_row.data_ptr(), pair2out=self.pair2out.data_ptr(),
                        pair2u=self.pair2u.data_ptr(), scratch=self.geo_scratch.data_ptr())

    def gemm_mode(self):
        """'h2' (split-f16 MFMA) or 'f32' for the next forward / sampling call on this batch"""
        mode = self.gemm or OPTIONS.gemm
        return mode if (mode == "h2" and self.weights16 is not None and self.bucket_weights16 is not None) else "f32"

    def struct(self):
        h2 = self.gemm_mode() == "h2"
        return Batch(
            num_nodes=self.N, num_graphs=self.G, num_pairs=self.P, num_models=self.M,
            graph_ptr=self.graph_ptr.data_ptr(), node_graph=self.node_graph.data_ptr(),
            pair_ptr=self.pair_ptr.data_ptr(), pair_code=self.pair_code.data_ptr(),
            weights=self.weights.data_ptr(), z=self.z.data_ptr(), x1_0=self.x1_0.data_ptr(),
            geo=self.geo_struct(), workspace=self.workspace.data_ptr(),
            edge_inv_u=self.edge_inv_u.data_ptr(),
            # 0 = "unknown": the library then runs the step tail as three launches (A/B switch for tests / tools)
            max_graph_nodes=self.max_n if (OPTIONS.fused_step_tail and not self.per_block) else 0,
            reserved=self.reserved_flags(),
            enc_tiles=self._tiles_struct("enc"), diff_tiles=self._tiles_struct("diff"),
            bucket_weights=None if self.bucket_weights is None else self.bucket_weights.data_ptr(),
            weights16=self.weights16.data_ptr() if h2 else None,
            bucket_weights16=self.bucket_weights16.data_ptr() if h2 else None,
            status=self.status.data_ptr(),
            unit_node=None if self.unit_node is None else self.unit_node.data_ptr(),
            num_units=0 if self.unit_node is None else int(self.unit_node.numel()) - 1,
            reserved2=0)

    def _tiles_struct(self, which):
        t = self.typed
        if t is None or self.bucket_weights is None or not OPTIONS.typed_tiles:
            return _lib.TypedTiles()
        e = which == "enc"
        tile, cap = (t["enc_tile"] if e else t["diff_tile"]), t["cap"]
        v = t["v"][0:3] if e else t["v"][3:6]
        return _lib.TypedTiles(num_tiles=t["nt_e"] if e else t["nt_d"], num_buckets=t["nb_e"] if e else t["nb_d"],
                               tile_slot=tile.data_ptr(), tile_start=tile[cap:].data_ptr(),
                               tile_count=tile[2 * cap:].data_ptr(), pair=v[0].data_ptr(), node_i=v[1].data_ptr(),
                               node_j=v[2].data_ptr())

    def train_struct(self, h2=False):
        """tsd_batch for the training step: topology + edge-list buffers only (no bound checkpoints).  h2: the step's
        tile GEMMs on the f16 MFMA pipes (tsd_batch.reserved bit 5), range flag in the batch's status word"""
        return Batch(
            reserved=(32 | (0 if OPTIONS.train_side_lane else 64)) if h2 else 0, status=self.status.data_ptr(),
            num_nodes=self.N, num_graphs=self.G, num_pairs=self.P, num_models=0,
            graph_ptr=self.graph_ptr.data_ptr(), node_graph=self.node_graph.data_ptr(),
            pair_ptr=self.pair_ptr.data_ptr(), pair_code=self.pair_code.data_ptr(),
            weights=None, z=None, x1_0=None, geo=self.geo_struct(), workspace=None, edge_inv_u=None)

    # ---- ops -------------------------------------------------------------------------------
    def geometry(self, pos):
        lib = _lib.load()
        self.check_status()
        self.geo_gen += 1
        pos = pos.to(torch.float32).contiguous()
        check(lib.tsd_geometry_build(C.byref(self.cfg), self.N, self.G, self.P, ptr(pos), ptr(self.graph_ptr),
                                     ptr(self.node_graph), ptr(self.pair_ptr), ptr(self.pair_code),
                                     self.geo_struct(), stream_ptr()))

    def build_geometry_async(self, pos):
        """the edge lists of `pos` on the current stream WITHOUT the host read of the topology status that geometry()
        makes first (prefetch_batch: the status word travels with the counts, counts_to_host_async)"""
        lib = _lib.load()
        self.geo_gen += 1
        check(lib.tsd_geometry_build(C.byref(self.cfg), self.N, self.G, self.P, ptr(pos), ptr(self.graph_ptr),
                                     ptr(self.node_graph), ptr(self.pair_ptr), ptr(self.pair_code),
                                     self.geo_struct(), stream_ptr()))

    def counts_to_host_async(self, host):
        """[enc_u count, out_u count, topology status, diff_u count] -> the pinned host tensor `host` (int32 [4]), written by
        one small kernel on the current stream (no host wait here; the caller records an event behind it) -- the layout of
        tsd_train_forward's counts_host"""
        check(_lib.load().tsd_geometry_counts_async(self.geo_struct(), ptr(self.status), ptr(host), stream_ptr()))
        return host

    def forward(self, pos):
        """geometry + M forwards; results stay on the device (self.edge_inv_u[m, :E_out/2])."""
        lib = _lib.load()
        self.check_status()
        self.geo_gen += 1
        pos = pos.to(torch.float32).contiguous()
        b = self.struct()
        check(lib.tsd_score_forward(C.byref(self.cfg), C.byref(b), ptr(pos), stream_ptr()))
        return pos

    def reserved_flags(self):
        """tsd_batch.reserved (include/tsdiff_hip.h): bit 0 one launch per block, bit 1 narrow filter tiles, bit 2 no
        fused encoder; test bits"""
        # the fused per-unit encoder: on by itself where it wins -- every unit one graph of more than 32 atoms (8 x 8
        # atom-block tiles; BASELINE configs[4]: -12 % step time) -- and off for small-molecule batches, where the
        # materialised forms keep two workgroups per CU busy and measure faster; "force": everywhere
        fe = OPTIONS.fused_encoder
        fused_on = fe == "force" or (bool(fe) and self.units_single_graph and not self.per_block)
        return ((0 if (OPTIONS.one_launch and not self.per_block) else 1) | (0 if OPTIONS.wide_filter_tiles else 2) |
                (0 if fused_on else 4) | (16 if fe == "force" else 0) | self.test_flags)

    def forward_out_edges(self, pos):
        """forward(pos) and the number of directed out edges, in ONE host read (the edge count and the status word).
        A split-f16 call that left the f16 range is rerun on the fp32-MFMA kernels; a one-launch forward in which a
        bounded in-kernel wait gave up (TSD_STATUS_INTERNAL: e.g. another tenant held the GPU's workgroup slots) is
        rerun as one launch per block -- bit-identical, no in-kernel waits."""
        for _ in range(3):
            self.forward(pos)
            words = torch.stack([self.out.count[0], self.status[0]]).cpu()
            if not self.status_fallback(int(words[1])):
                return int(words[0])
        raise _lib.TsdError("internal: the forward kept reporting status %d" % int(words[1]))

    def status_fallback(self, word):
        """The status word of a finished forward / sampling call: False if the results stand.  True: the batch has been
        switched to the form that cannot fail that way and the caller reruns the call -- TSD_STATUS_RANGE (split-f16
        arithmetic left the f16 range) -> fp32-MFMA kernels; TSD_STATUS_INTERNAL (a bounded wait inside the one-launch
        forward or the fused step tail gave up) -> one launch per block / three-launch tail.  The bits are cleared."""
        word = int(word)
        if not (word & (_lib.STATUS_RANGE | _lib.STATUS_INTERNAL)):
            return False
        self.status[:1].zero_()
        if word & _lib.STATUS_INTERNAL:
            if self.per_block:
                raise _lib.TsdError("internal: a bounded in-kernel wait gave up (TSD_STATUS_INTERNAL) in the "
                                    "launch-per-block form")
            self.per_block = True
            self.drop_plans()
        if word & _lib.STATUS_RANGE:
            if self.gemm_mode() != "h2":
                raise _lib.TsdError("internal: TSD_STATUS_RANGE from an fp32 forward")
            import warnings
            warnings.warn("tsdiff_amd: an activation left the range the split-f16 arithmetic covers (|a| > 65504, or a tile "
                          "of operands entirely below 2^-12); this batch now runs on the fp32-input MFMA kernels (same "
                          "results, about half the speed).  model.preflight_split_f16() reports how the weights sit.",
                          RuntimeWarning, stacklevel=3)
            self.gemm = "f32"
        return True

    def range_fallback(self, word):
        """(kept for callers of the round-3 name) see status_fallback"""
        return self.status_fallback(word)

    def ensemble_mean(self):
        """directed edge_inv (reference order): mean over the checkpoints, expanded through out.umap"""
        lib = _lib.load()
        mean = self.scratch[: max(self.P, 1)]
        check(lib.tsd_ensemble_mean(self.M, self.P, self.out.struct(), ptr(self.edge_inv_u), ptr(mean),
                                    stream_ptr()))
        return mean

    def eq_transform_rows(self, pos, score_d):
        lib = _lib.load()
        score = torch.empty(self.N, 3, dtype=torch.float32, device=self.device)
        check(lib.tsd_eq_transform_rows(self.N, ptr(pos), ptr(self.pair_ptr), ptr(self.graph_ptr),
                                        ptr(self.node_graph), self.out.struct(), ptr(self.pair2out),
                                        ptr(score_d), ptr(score), stream_ptr()))
        return score

    def edges_to_torch(self, which="out"):
        """(edge_index (2,E) int64, edge_length (E,1), type_r, type_p) -- syncs to read E."""
        el = self.out if which == "out" else self.enc
        E = el.num_edges()
        ei = torch.stack([el.src[:E], el.dst[:E]]).to(torch.int64)
        return ei, el.dist[:E].clone().unsqueeze(-1), el.type_r[:E].to(torch.int64), el.type_p[:E].to(torch.int64)

    # ---- the device-resident sampling loop ---------------------------------------------------
    def drop_plans(self):
        if not getattr(self, "_plans", None):
            return
        for s in getattr(self, "_plan_streams", {}).values():  # an exec must outlive its launches: wait for the
            s.synchronize()                                     # streams the plans ran on (not for the whole device)
        lib = _lib.load()
        for plan in self._plans.values():
            lib.tsd_sampler_plan_destroy(plan)
        self._plans = {}
        self._plan_streams = {}

    def __del__(self):
        try:
            if sys is None or sys.is_finalizing():  # interpreter shutdown: the driver reclaims everything, no sync from GC
                return
            self.drop_plans()
        except Exception:
            pass

    def sampler_plan(self, kind, clip, clip_pos):
        """the captured + instantiated hipGraph of one sampling step for the bound checkpoints; built once and
        replayed by every later dynamic_sampling call on this batch (reference loop: models/sampler.py:187-254)"""
        key = (int(kind), float(clip), float(-1.0 if clip_pos is None else clip_pos),
               bool(OPTIONS.fused_step_tail and not self.per_block),
               bool(OPTIONS.typed_tiles), self.gemm_mode(), self.reserved_flags())
        plan = self._plans.get(key)
        if plan is None:
            lib = _lib.load()
            self.check_status()
            b = self.struct()
            side = _side_stream(self.device)  # capture is illegal on the legacy default stream
            side.wait_stream(torch.cuda.current_stream(self.device))
            plan = C.c_void_p()
            check(lib.tsd_sampler_plan_create(C.byref(self.cfg), C.byref(b), key[0], key[1], key[2],
                                              ptr(self.pos_work), ptr(self.status), C.c_void_p(side.cuda_stream),
                                              C.byref(plan)))
            self._plans[key] = plan
        return plan

    def sampler_run(self, kind, coefs, noises, seed, offset, clip, clip_pos, traj, use_graph=True):
        """n = coefs.shape[0] steps on self.pos_work, asynchronous on the current stream (no host sync).
        noises None: device Philox draws keyed by (seed, offset)."""
        lib = _lib.load()
        self.geo_gen += 1
        cur = torch.cuda.current_stream(self.device)
        if not use_graph:
            # eager (debugging) form: the same kernels launched one by one, NO capture and no plan object -- usable
            # when capture itself is what is being diagnosed (tsd_sampler_run's eager branch does not synchronise)
            self.check_status()
            b = self.struct()
            check(lib.tsd_sampler_run(C.byref(self.cfg), C.byref(b), int(kind), int(coefs.shape[0]), ptr(coefs),
                                      ptr(noises), int(seed), int(offset), float(clip),
                                      float(-1.0 if clip_pos is None else clip_pos), ptr(self.pos_work), ptr(traj),
                                      ptr(self.status), 0, C.c_void_p(cur.cuda_stream)))
            return
        plan = self.sampler_plan(kind, clip, clip_pos)
        args = _lib.RunArgs(coefs=coefs.data_ptr(), noises=None if noises is None else noises.data_ptr(),
                            traj=None if traj is None else traj.data_ptr(), seed=int(seed), offset=int(offset))
        # a graph may not be launched into the legacy default stream either: hop to the side stream
        side = _side_stream(self.device) if cur.cuda_stream == 0 else None
        run_on = side if side is not None else cur
        if side is not None:
            side.wait_stream(cur)
        self._plan_streams[run_on.cuda_stream] = run_on
        check(lib.tsd_sampler_plan_run(plan, int(coefs.shape[0]), C.byref(args), 1, C.c_void_p(run_on.cuda_stream)))
        if side is not None:
            cur.wait_stream(side)


def partition_units(nodes_per_graph, num_models, num_cus, max_nodes=_lib.UNIT_MAX_NODES):
    """Node offsets [U + 1] of the units of the fused per-unit encoder (csrc/kernels_unit.hip): consecutive runs of
    whole graphs, at most `max_nodes` atoms each.  One workgroup owns a unit on a CU of its own, its time is ~ the number
    of 64-pair filter tiles of the unit, and the U x M workgroups of a launch are dealt to the CUs in order; so the runs
    are cut at a cap on the pair count, and the cap is the one of a few candidates (1, 2, 3, 4, 6, 8 rounds of
    workgroups per CU) with the smallest simulated makespan.  Pure host arithmetic on the graph sizes."""
    n = np.asarray(nodes_per_graph, dtype=np.int64)
    pairs = n * (n - 1) // 2
    G = int(n.shape[0])
    total = int(pairs.sum())

    def cut(cap):
        bounds, atoms, acc = [0], 0, 0
        node = 0
        for g in range(G):
            if atoms > 0 and (atoms + n[g] > max_nodes or acc + pairs[g] > cap):
                bounds.append(node)
                atoms, acc = 0, 0
            atoms += int(n[g])
            acc += int(pairs[g])
            node += int(n[g])
        bounds.append(node)
        return np.asarray(bounds, dtype=np.int64)

    def makespan(bounds):
        # tiles per unit (+ a node-chain term), workgroups dealt to the least loaded CU in dispatch order
        first = np.searchsorted(np.cumsum(n), bounds[1:], side="left")  # graphs per unit via node offsets
        gsum = np.concatenate([[0], np.cumsum(pairs)])
        gi = np.concatenate([[0], first + 1])
        gi = np.minimum(gi, G)
        up = gsum[gi[1:]] - gsum[gi[:-1]]
        cost = np.ceil(up / 64.0) + 1.0
        cost = np.tile(cost, num_models)
        load = np.zeros(num_cus)
        for c in cost:
            k = int(np.argmin(load))
            load[k] += c
        return float(load.max())

    best = None
    for rounds in (1, 2, 3, 4, 6, 8):
        units_wanted = max(1, (num_cus * rounds) // max(num_models, 1))
        cap = max(int(np.ceil(total / units_wanted)), int(pairs.max()) if G else 1, 1)
        b = cut(cap)
        ms = makespan(b) if len(b) - 1 <= 4096 else float(len(b))
        if best is None or ms < best[0] - 1e-9:
            best = (ms, b)
        if len(b) - 1 >= G:
            break
    return best[1]


def eq_transform(score_d, pos, edge_index, edge_length):
    """reference models/geometry.py:22-30 on arbitrary (device) edge lists."""
    lib = _lib.load()
    if pos.device.type != "cuda":
        raise _lib.TsdError("tsdiff_amd.eq_transform needs cuda tensors (no CPU fallback)")
    N = int(pos.shape[0])
    E = int(edge_index.shape[1])
    score = torch.zeros(N, 3, dtype=torch.float32, device=pos.device)
    sd = score_d.detach().to(torch.float32).contiguous().view(-1)
    ln = edge_length.detach().to(torch.float32).contiguous().view(-1)
    ei = edge_index.to(torch.int64).contiguous()
    p = pos.detach().to(torch.float32).contiguous()
    check(lib.tsd_eq_transform(N, E, ptr(sd), ptr(p), ptr(ei), ptr(ln), ptr(score), stream_ptr()))
    return score
