"""ctypes binding of libtsdiff_hip.so (include/tsdiff_hip.h).

The product path has NO fallback: if the HIP library is missing or a call fails, an exception is
raised.  Build it with `python __graft_entry__.py` (or `make -C tsdiff_amd/csrc`).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtsdiff_hip.so")

TSD_OK = 0
TSD_ERR_INVALID = -1
TSD_ERR_HIP = -2
TSD_ERR_UNSUPPORTED = -3
TSD_ERR_NAN = -4
TSD_ERR_RANGE = -5

EDGE_TILE = 32
EDGE_PAD = 8  # TSD_EDGE_PAD: spare entries every tsd_edges array carries past its capacity
MAX_GRAPH_NODES = 255
UNIT_MAX_NODES = 64  # TSD_UNIT_MAX_NODES
STEP_COEFS = 8
STATUS_NAN = 1
STATUS_BAD_BOND = 2
STATUS_ASYMMETRIC = 4
STATUS_INTERNAL = 8
STATUS_RANGE = 16  # split-f16 forward: an activation left the f16 range (the call is rerun on the fp32-MFMA kernels)

c_i32p = C.POINTER(C.c_int32)
c_f32p = C.POINTER(C.c_float)


class ModelCfg(C.Structure):
    _fields_ = [
        ("hidden", C.c_int32),
        ("num_convs", C.c_int32),
        ("feat_dim", C.c_int32),
        ("edge_order", C.c_int32),
        ("pred_edge_order", C.c_int32),
        ("edge_cutoff", C.c_float),
        ("conv_cutoff", C.c_float),
        ("smooth_conv", C.c_int32),
    ]


class Edges(C.Structure):
    _fields_ = [
        ("count", C.c_void_p),
        ("row_ptr", C.c_void_p),
        ("src", C.c_void_p),
        ("dst", C.c_void_p),
        ("dist", C.c_void_p),
        ("type_r", C.c_void_p),
        ("type_p", C.c_void_p),
        ("pair_id", C.c_void_p),
        ("umap", C.c_void_p),
    ]


class Geometry(C.Structure):
    _fields_ = [
        ("enc", Edges),
        ("out", Edges),
        ("enc_u", Edges),
        ("out_u", Edges),
        ("diff_u", Edges),
        ("attr_row", C.c_void_p),
        ("pair2out", C.c_void_p),
        ("pair2u", C.c_void_p),
        ("scratch", C.c_void_p),
    ]


class TypedTiles(C.Structure):  # tsd_typed_tiles
    _fields_ = [
        ("num_tiles", C.c_int32),
        ("num_buckets", C.c_int32),
        ("tile_slot", C.c_void_p),
        ("tile_start", C.c_void_p),
        ("tile_count", C.c_void_p),
        ("pair", C.c_void_p),
        ("node_i", C.c_void_p),
        ("node_j", C.c_void_p),
    ]


class Batch(C.Structure):
    _fields_ = [
        ("num_nodes", C.c_int32),
        ("num_graphs", C.c_int32),
        ("num_pairs", C.c_int32),
        ("num_models", C.c_int32),
        ("graph_ptr", C.c_void_p),
        ("node_graph", C.c_void_p),
        ("pair_ptr", C.c_void_p),
        ("pair_code", C.c_void_p),
        ("weights", C.c_void_p),
        ("z", C.c_void_p),
        ("x1_0", C.c_void_p),
        ("geo", Geometry),
        ("workspace", C.c_void_p),
        ("edge_inv_u", C.c_void_p),
        ("max_graph_nodes", C.c_int32),
        ("reserved", C.c_int32),
        ("enc_tiles", TypedTiles),
        ("diff_tiles", TypedTiles),
        ("bucket_weights", C.c_void_p),
        ("weights16", C.c_void_p),          # appended in 0.4: f16-plane arenas of the split-f16 forward, range word
        ("bucket_weights16", C.c_void_p),
        ("status", C.c_void_p),
        ("unit_node", C.c_void_p),          # appended in 0.5: unit partition of the fused per-unit encoder
        ("num_units", C.c_int32),
        ("reserved2", C.c_int32),
    ]


class Work(C.Structure):
    """tsd_work"""
    _fields_ = [(k, C.c_double) for k in ("flops_edge_embed", "flops_blocks", "flops_pair_output", "flops_other",
                                          "flops_executed", "flops_reference", "flops_block_launch", "bytes_aggregate",
                                          "flops_train_forward")]


class RunArgs(C.Structure):  # tsd_run_args
    _fields_ = [
        ("coefs", C.c_void_p),
        ("noises", C.c_void_p),
        ("traj", C.c_void_p),
        ("seed", C.c_uint64),
        ("offset", C.c_uint64),
    ]


SAMPLER_STATE_INTS = 16  # sizeof(tsd_sampler_state) / 4; [0] = flags, [1] = step counter


# name -> (restype, argtypes); every symbol declared in include/tsdiff_hip.h
_P = C.c_void_p
_CFG = C.POINTER(ModelCfg)
SIGNATURES = {
    "tsd_version": (C.c_char_p, []),
    "tsd_last_error": (C.c_char_p, []),
    "tsd_raw_weight_floats": (C.c_size_t, [_CFG]),
    "tsd_packed_weight_floats": (C.c_size_t, [_CFG]),
    "tsd_pack_weights": (C.c_int, [_CFG, _P, _P, _P]),
    "tsd_topology_build": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int64, _P, _P, _P, _P, C.c_int32,
                                     C.c_int32, _P, _P, _P, _P, _P]),
    "tsd_geometry_scratch_ints": (C.c_size_t, [C.c_int32, C.c_int32]),
    "tsd_geometry_build": (C.c_int, [_CFG, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, Geometry, _P]),
    "tsd_node_embed": (C.c_int, [_CFG, _P, C.c_int32, _P, _P, _P, _P, _P]),
    "tsd_edge_embed": (C.c_int, [_CFG, _P, C.c_int32, Edges, _P, _P]),
    "tsd_node_lin1": (C.c_int, [_CFG, _P, C.c_int32, C.c_int32, _P, _P, _P]),
    "tsd_cfconv_layer": (C.c_int, [_CFG, _P, C.c_int32, C.c_int32, Edges, _P, _P, _P, _P, _P]),
    "tsd_filter_gen": (C.c_int, [_CFG, _P, C.c_int32, Edges, _P, _P, _P]),
    "tsd_interaction_block": (C.c_int, [_CFG, _P, C.c_int32, C.c_int32, Edges, _P, _P, _P, _P, C.c_int32, C.c_int32,
                                        Edges, _P, _P, _P]),
    "tsd_attr_planes": (C.c_int, [C.c_int32, C.c_int64, _P, _P, _P, _P]),
    "tsd_interaction_block16": (C.c_int, [_CFG, _P, C.c_int32, C.c_int32, Edges, _P, _P, _P, _P, C.c_int32, C.c_int32,
                                          Edges, _P, _P, _P, _P]),
    "tsd_cfconv_aggregate": (C.c_int, [C.c_int32, C.c_int32, _P, _P, _P, _P, _P, _P, _P]),
    "tsd_node_update": (C.c_int, [_CFG, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, _P]),
    "tsd_pair_output": (C.c_int, [_CFG, _P, C.c_int32, Edges, _P, _P, _P, _P, _P]),
    "tsd_eq_transform": (C.c_int, [C.c_int32, C.c_int64, _P, _P, _P, _P, _P, _P]),
    "tsd_forward_workspace_floats": (C.c_size_t, [_CFG, C.c_int32, C.c_int32, C.c_int32]),
    "tsd_score_forward": (C.c_int, [_CFG, C.POINTER(Batch), _P, _P]),
    "tsd_ensemble_mean": (C.c_int, [C.c_int32, C.c_int32, Edges, _P, _P, _P]),
    "tsd_eq_transform_rows": (C.c_int, [C.c_int32, _P, _P, _P, _P, Edges, _P, _P, _P, _P]),
    "tsd_sampler_step": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, C.c_float, C.c_float, _P,
                                   _P, _P]),
    "tsd_linear_fwd": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "tsd_linear_bwd": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "tsd_linear_packable": (C.c_int, [C.c_int32, C.c_int32]),
    "tsd_pack_linear_batch": (C.c_int, [C.c_int32, _P, _P, _P, _P, _P, _P]),
    "tsd_linear_fwd_packed": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P]),
    "tsd_act_fwd": (C.c_int, [C.c_int32, C.c_int64, _P, _P, _P]),
    "tsd_act_bwd": (C.c_int, [C.c_int32, C.c_int64, _P, _P, _P, _P]),
    "tsd_emb_mul_fwd": (C.c_int, [C.c_int32, C.c_int32, _P, _P, _P, _P, _P]),
    "tsd_emb_mul_bwd": (C.c_int, [C.c_int32, C.c_int32, _P, _P, _P, _P, _P, _P, _P]),
    "tsd_gather_rows": (C.c_int, [C.c_int32, C.c_int32, _P, _P, _P, _P]),
    "tsd_scatter_rows_add": (C.c_int, [C.c_int32, C.c_int32, _P, _P, _P, _P]),
    "tsd_row_mask": (C.c_int, [C.c_int32, C.c_int32, _P, C.c_float, C.c_int32, _P, _P]),
    "tsd_aggregate_bwd_filter": (C.c_int, [C.c_int32, C.c_int32, Edges, _P, _P, _P, _P]),
    "tsd_pair_product_fwd": (C.c_int, [C.c_int32, C.c_int32, Edges, _P, _P, _P]),
    "tsd_pair_product_bwd": (C.c_int, [C.c_int32, C.c_int32, Edges, _P, _P, _P, _P]),
    "tsd_eq_und_fwd": (C.c_int, [C.c_int32, Edges, _P, _P, _P, _P]),
    "tsd_eq_und_bwd": (C.c_int, [C.c_int32, Edges, _P, _P, _P, _P]),
    "tsd_pair_distance": (C.c_int, [C.c_int32, Edges, _P, _P, _P]),
    "tsd_gine_aggregate": (C.c_int, [C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_float, _P, _P, _P, _P, _P]),
    "tsd_gaussian_edge_encode": (C.c_int, [C.c_int64, C.c_int32, C.c_float, _P, _P, _P, _P, _P, _P]),
    "tsd_diffuse_positions": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, _P, _P, _P]),
    "tsd_geometry_counts_async": (C.c_int, [Geometry, _P, _P, _P]),
    "tsd_train_raw_floats": (C.c_size_t, [_CFG]),
    "tsd_train_workspace_floats": (C.c_size_t, [_CFG, C.c_int32, C.c_int32]),
    "tsd_train_forward": (C.c_int, [_CFG, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_size_t, _P, _P, _P]),
    "tsd_train_backward": (C.c_int, [_CFG, _P, _P, _P, _P, _P, C.c_size_t, _P, _P, _P, _P]),
    "tsd_train_backward2": (C.c_int, [_CFG, _P, _P, _P, _P, _P, C.c_size_t, _P, _P, _P, _P, _P]),
    "tsd_train_grad_buckets": (C.c_int, [_CFG, C.POINTER(C.c_size_t)]),
    "tsd_forward_work": (C.c_int, [_CFG, C.c_int32, C.c_int64, C.c_int64, C.c_int64, C.POINTER(Work)]),
    "tsd_grad_norm_clip": (C.c_int, [C.c_int64, _P, C.c_float, _P, _P, _P]),
    "tsd_adam_step": (C.c_int, [C.c_int64, _P, _P, _P, _P, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                C.c_int64, _P]),
    "tsd_gine_csr_fwd": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_float, Edges, _P, _P, _P, _P]),
    "tsd_gine_csr_bwd": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, Edges, Edges, _P, _P, _P,
                                   _P, _P, _P]),
    "tsd_embedding_renorm": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _P, C.c_float, _P, _P, _P]),
    "tsd_dual_score": (C.c_int, [C.c_int32, _P, _P, C.c_float, C.c_float, C.c_float, _P, _P]),
    "tsd_sampler_plan_create": (C.c_int, [_CFG, C.POINTER(Batch), C.c_int32, C.c_float, C.c_float, _P, _P, _P,
                                          C.POINTER(_P)]),
    "tsd_sampler_plan_run": (C.c_int, [_P, C.c_int32, C.POINTER(RunArgs), C.c_int32, _P]),
    "tsd_sampler_plan_destroy": (None, [_P]),
    "tsd_sampler_run": (C.c_int, [_CFG, C.POINTER(Batch), C.c_int32, C.c_int32, _P, _P, C.c_uint64, C.c_uint64,
                                  C.c_float, C.c_float, _P, _P, _P, C.c_int32, _P]),
    "tsd_philox_normal": (C.c_int, [C.c_uint64, C.c_uint64, C.c_int64, _P, _P]),
    "tsd_typed_tiles_capacity": (C.c_size_t, [C.c_int32]),
    "tsd_typed_tiles_build": (C.c_int, [_CFG, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                        _P, _P, _P]),
    "tsd_bucket_weights_floats": (C.c_size_t, [_CFG, C.c_int32]),
    "tsd_bucket_weights_build": (C.c_int, [_CFG, _P, C.c_int32, _P, _P, _P]),
    "tsd_forward_blocks": (C.c_int, [_CFG, C.POINTER(Batch), C.c_int32, _P]),
    "tsd_forward_encoder": (C.c_int, [_CFG, C.POINTER(Batch), C.c_int32, C.c_int32, _P]),
    "tsd_weights16_preflight": (C.c_int, [_P, C.c_size_t, _P, _P]),
    "tsd_forward_workspace_layout": (C.c_int, [_CFG, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_size_t)]),
    "tsd_pack_weights16": (C.c_int, [_CFG, _P, _P, _P]),
    "tsd_bucket_weights16": (C.c_int, [_CFG, _P, C.c_int32, _P, _P]),
}

_lib = None


class TsdError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle; raises if the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TsdError(
            f"{LIB_PATH} not found: the HIP extension is not built (run `python __graft_entry__.py` "
            "or `make -C tsdiff_amd/csrc`). There is no CPU fallback for the product path.")
    lib = C.CDLL(LIB_PATH)
    variant = os.path.abspath(LIB_PATH) != os.path.join(_HERE, "libtsdiff_hip.so")
    for name, (res, args) in SIGNATURES.items():
        if variant and not hasattr(lib, name):
            continue  # (tools/ab_*.py with an OLDER library, e.g. last round's for a same-box A/B: entries it lacks stay unbound)
        fn = getattr(lib, name)  # AttributeError if the .so is stale -> loud
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code):
    """Map a TSD_* return code to the reference's error conventions."""
    if code == TSD_OK:
        return
    msg = load().tsd_last_error().decode(errors="replace")
    if code == TSD_ERR_INVALID:
        raise ValueError(msg)
    if code == TSD_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    if code == TSD_ERR_NAN:
        raise FloatingPointError(msg)
    raise TsdError(msg)


def ptr(t):
    """device pointer of a torch tensor (or None) as c_void_p"""
    if t is None:
        return C.c_void_p(0)
    return C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
