"""Deterministic synthetic inputs for parity tests and bench.py.

Trained TSDiff checkpoints and the wb97xd3 pickles are LFS blobs that are absent
from the reference tree (reference .MISSING_LARGE_BLOBS:1-12), so every workload
here is generated: weights by a counter-based hash (bit-stable across numpy
versions and machines, no 11 MB blob to ship), graphs by the recipe of
SURVEY.md section 8(d).

Shapes / names of the weight tensors follow the reference `state_dict`
(reference models/epsnet/condensenc.py:48-115, models/encoder/schnet.py:74-171,
models/encoder/edge.py:44-56).
"""
import numpy as np

NUM_BOND_TYPES = 22  # len(rdkit BondType.names), reference utils/chem.py:21
FEAT_BLOCKS = (2, 3, 4, 3, 4, 4, 3, 2)  # data/TS/wb97xd3/feat_dict.pkl -> feat_dim 25

DEFAULT_MODEL_CONFIG = {
    # reference configs/train_config.yml:1-32
    "type": "diffusion",
    "network": "condensenc",
    "t0": 0,
    "t1": 5000,
    "edge_cutoff": 10.0,
    "edge_order": 4,
    "pred_edge_order": 3,
    "encoder": {
        "name": "schnet",
        "edge_emb": False,
        "num_convs": 7,
        "cutoff": 10.0,
        "smooth_conv": False,
        "mlp_act": "swish",
        "hidden_dim": 256,
    },
    "feat_dim": 25,
    "hidden_dim": 256,
    "edge_encoder": "mlp",
    "mlp_act": "swish",
    "edge_cat_act": "swish",
    "beta_schedule": "sigmoid",
    "beta_start": 1.0e-7,
    "beta_end": 2.0e-3,
    "num_diffusion_timesteps": 5000,
}


def small_model_config(hidden=64, num_convs=2):
    import copy
    cfg = copy.deepcopy(DEFAULT_MODEL_CONFIG)
    cfg["hidden_dim"] = hidden
    cfg["encoder"]["hidden_dim"] = hidden
    cfg["encoder"]["num_convs"] = num_convs
    return cfg


# ----------------------------------------------------------------------------
# counter-based uniform generator (splitmix64 finaliser)
# ----------------------------------------------------------------------------
def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def hash_uniform(n, seed, stream=0):
    """n float64 values uniform in [0, 1), function of (seed, stream, index) only."""
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        key = _splitmix64(np.uint64(seed) * np.uint64(0x632BE59BD9B4E019)
                          + np.uint64(stream) * np.uint64(0xD1342543DE82EF95))
        bits = _splitmix64(idx ^ key)
    return (bits >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def _stream_id(name):
    h = np.uint64(1469598103934665603)
    with np.errstate(over="ignore"):
        for ch in name.encode():
            h = (h ^ np.uint64(ch)) * np.uint64(1099511628211)
    return int(h & np.uint64(0x7FFFFFFF))


def param_shapes(cfg):
    """name -> (shape, fan_in) for every *trainable* tensor of the reference model."""
    H = int(cfg["hidden_dim"])
    enc = cfg["encoder"]
    He = int(enc["hidden_dim"])
    L = int(enc["num_convs"])
    F = int(cfg["feat_dim"])
    assert H == He, "condensenc feeds hidden_dim-wide z into the encoder"
    shapes = {
        "edge_encoder.bond_emb.weight": ((100, H), None),
        "edge_encoder.mlp.layers.0.weight": ((H, 1), 1),
        "edge_encoder.mlp.layers.0.bias": ((H,), 1),
        "edge_encoder.mlp.layers.1.weight": ((H, H), H),
        "edge_encoder.mlp.layers.1.bias": ((H,), H),
        "atom_embedding.weight": ((100, H // 2), None),
        "atom_feat_embedding.weight": ((H // 2, F), F),
    }
    for l in range(L):
        p = f"encoder.interactions.{l}."
        shapes[p + "conv.lin1.weight"] = ((He, He), He)
        shapes[p + "conv.lin2.weight"] = ((He, He), He)
        shapes[p + "conv.lin2.bias"] = ((He,), He)
        shapes[p + "conv.nn.0.weight"] = ((He, He), He)
        shapes[p + "conv.nn.0.bias"] = ((He,), He)
        shapes[p + "conv.nn.2.weight"] = ((He, He), He)
        shapes[p + "conv.nn.2.bias"] = ((He,), He)
        shapes[p + "lin.weight"] = ((He, He), He)
        shapes[p + "lin.bias"] = ((He,), He)
    shapes["grad_dist_mlp.layers.0.weight"] = ((H, 2 * H), 2 * H)
    shapes["grad_dist_mlp.layers.0.bias"] = ((H,), 2 * H)
    shapes["grad_dist_mlp.layers.1.weight"] = ((H // 2, H), H)
    shapes["grad_dist_mlp.layers.1.bias"] = ((H // 2,), H)
    shapes["grad_dist_mlp.layers.2.weight"] = ((1, H // 2), H // 2)
    shapes["grad_dist_mlp.layers.2.bias"] = ((1,), H // 2)
    shapes["edge_cat.0.weight"] = ((H, 2 * H), 2 * H)
    shapes["edge_cat.0.bias"] = ((H,), 2 * H)
    shapes["edge_cat.2.weight"] = ((H, H), H)
    shapes["edge_cat.2.bias"] = ((H,), H)
    return shapes


def synth_state_dict(cfg, seed=0):
    """Closed-form weights with torch-default-like magnitudes.

    Linear weight/bias ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in)) (torch.nn.Linear
    default), embeddings ~ U(-sqrt(3), sqrt(3)) (unit variance like N(0,1)).
    Returns name -> float32 ndarray for the trainable tensors only; `betas` /
    `alphas` are derived from the config by the model itself.
    """
    out = {}
    for name, (shape, fan_in) in param_shapes(cfg).items():
        n = int(np.prod(shape))
        u = hash_uniform(n, seed, _stream_id(name))
        bound = np.sqrt(3.0) if fan_in is None else 1.0 / np.sqrt(fan_in)
        out[name] = ((2.0 * u - 1.0) * bound).astype(np.float32).reshape(shape)
    return out


# ----------------------------------------------------------------------------
# graphs
# ----------------------------------------------------------------------------
def _one_hot_feat(rng, n):
    cols = []
    for b in FEAT_BLOCKS:
        k = rng.integers(0, b, size=n)
        oh = np.zeros((n, b), dtype=np.int64)
        oh[np.arange(n), k] = 1
        cols.append(oh)
    return np.concatenate(cols, axis=1)


def _random_bond_graph(rng, n):
    """Spanning tree + 0-1 ring closure, valence <= 4; bond orders in {1,2,3,12}."""
    deg = np.zeros(n, dtype=np.int64)
    bonds = {}
    order = rng.permutation(n)
    for k in range(1, n):
        a = order[k]
        cands = [order[j] for j in range(k) if deg[order[j]] < 4]
        b = cands[rng.integers(0, len(cands))]
        bonds[(min(a, b), max(a, b))] = int(rng.choice([1, 1, 1, 2, 3, 12]))
        deg[a] += 1
        deg[b] += 1
    if n >= 4 and rng.random() < 0.5:
        for _ in range(16):
            a, b = rng.integers(0, n, size=2)
            key = (min(a, b), max(a, b))
            if a != b and key not in bonds and deg[a] < 4 and deg[b] < 4:
                bonds[key] = 1
                deg[a] += 1
                deg[b] += 1
                break
    return bonds


def _reaction_graph(rng, n):
    r = _random_bond_graph(rng, n)
    p = dict(r)
    keys = list(p.keys())
    del p[keys[rng.integers(0, len(keys))]]  # break one bond
    for _ in range(32):  # form one bond
        a, b = rng.integers(0, n, size=2)
        key = (min(a, b), max(a, b))
        if a != b and key not in p and key not in r:
            p[key] = 1
            break
    pairs = sorted(set(r) | set(p))
    ei, et = [], []
    for (a, b) in pairs:
        t = r.get((a, b), 0) * NUM_BOND_TYPES + p.get((a, b), 0)
        ei += [(a, b), (b, a)]
        et += [t, t]
    ei = np.asarray(ei, dtype=np.int64).reshape(-1, 2)
    et = np.asarray(et, dtype=np.int64)
    perm = np.lexsort((ei[:, 1], ei[:, 0]))  # row-major, reference utils/datasets.py:495-498
    return ei[perm].T.copy(), et[perm]


def collate(graphs):
    """PyG Batch.from_data_list semantics for the fields the path uses."""
    out = {k: [] for k in ("atom_type", "r_feat", "p_feat", "pos", "bond_index", "bond_type", "batch")}
    off = 0
    nn = []
    for g, d in enumerate(graphs):
        n = d["atom_type"].shape[0]
        out["atom_type"].append(d["atom_type"])
        out["r_feat"].append(d["r_feat"])
        out["p_feat"].append(d["p_feat"])
        out["pos"].append(d["pos"])
        out["bond_index"].append(d["bond_index"] + off)
        out["bond_type"].append(d["bond_type"])
        out["batch"].append(np.full(n, g, dtype=np.int64))
        nn.append(n)
        off += n
    res = {k: np.concatenate(v, axis=(1 if k == "bond_index" else 0)) for k, v in out.items()}
    res["num_nodes_per_graph"] = np.asarray(nn, dtype=np.int64)
    res["num_graphs"] = len(graphs)
    return res


def wb97xd3_like_batch(num_graphs=100, seed=0, n_lo=8, n_hi=23):
    """Config C2/C3/C4 stand-in for wb97xd3 test_data.pkl."""
    rng = np.random.default_rng(seed)
    graphs = []
    for _ in range(num_graphs):
        n = int(rng.integers(n_lo, n_hi + 1))
        bi, bt = _reaction_graph(rng, n)
        graphs.append({
            "atom_type": rng.choice(np.asarray([1, 1, 1, 6, 6, 7, 8], dtype=np.int64), size=n),
            "r_feat": _one_hot_feat(rng, n),
            "p_feat": _one_hot_feat(rng, n),
            "pos": rng.standard_normal((n, 3)).astype(np.float32),
            "bond_index": bi,
            "bond_type": bt,
        })
    return collate(graphs)


def dense_stress_batch(num_graphs=1024, n=64, seed=0, box=5.5):
    """Config C5: n-atom graphs inside a `box` Angstrom cube (every pair within the
    10 A cutoff) -> complete intra-graph pair set, E = G*n*(n-1)."""
    rng = np.random.default_rng(seed)
    graphs = []
    for _ in range(num_graphs):
        bi, bt = _reaction_graph(rng, n)
        pos = (rng.random((n, 3)) * box).astype(np.float32)
        pos -= pos.mean(0, keepdims=True)
        graphs.append({
            "atom_type": rng.choice(np.asarray([1, 1, 1, 6, 6, 7, 8], dtype=np.int64), size=n),
            "r_feat": _one_hot_feat(rng, n),
            "p_feat": _one_hot_feat(rng, n),
            "pos": pos,
            "bond_index": bi,
            "bond_type": bt,
        })
    return collate(graphs)


def replicate(graph, times, pos_list=None):
    """Same reaction `times` times (sampling.py batches `repeat` copies)."""
    gs = []
    for k in range(times):
        d = dict(graph)
        if pos_list is not None:
            d["pos"] = np.asarray(pos_list[k], dtype=np.float32)
        gs.append(d)
    return collate(gs)


# ----------------------------------------------------------------------------
# GeoDiff legacy dual-encoder network (reference configs/geodiff_legacy/qm9_default.yml:1-16)
# ----------------------------------------------------------------------------
LEGACY_QM9_MODEL_CONFIG = {
    "type": "diffusion", "network": "dualenc", "hidden_dim": 128, "num_convs": 6, "num_convs_local": 4,
    "cutoff": 10.0, "mlp_act": "ReLU", "beta_schedule": "sigmoid", "beta_start": 1.0e-7, "beta_end": 2.0e-3,
    "num_diffusion_timesteps": 5000, "edge_order": 3, "edge_encoder": "mlp", "smooth_conv": False,
}


def small_dual_config(hidden=64, num_convs=2, num_convs_local=2, ts=False):
    cfg = dict(LEGACY_QM9_MODEL_CONFIG)
    cfg.update(hidden_dim=hidden, num_convs=num_convs, num_convs_local=num_convs_local)
    if ts:
        cfg.update(TS=True, edge_cat_act="ReLU")
    return cfg


def single_bond_types(bond_type):
    """composite r*22+p reaction bond types -> one plain bond type per bond (1..21), for the single-graph
    legacy network"""
    bt = np.asarray(bond_type)
    r, p = bt // NUM_BOND_TYPES, bt % NUM_BOND_TYPES
    return np.where(r > 0, r, p).astype(np.int64)


def hash_state_dict(named_shapes, seed=0):
    """closed-form weights for ANY list of (name, shape): embeddings ~ U(-sqrt3, sqrt3), matrices and
    biases ~ U(-1/sqrt(fan), 1/sqrt(fan)) with fan = last dimension (matrices) or length (vectors)."""
    out = {}
    for name, shape in named_shapes:
        shape = tuple(int(x) for x in shape)
        n = int(np.prod(shape)) if shape else 1
        u = hash_uniform(n, seed, _stream_id(name))
        bound = np.sqrt(3.0) if "emb." in name else 1.0 / np.sqrt(max(shape[-1] if shape else 1, 1))
        out[name] = ((2.0 * u - 1.0) * bound).astype(np.float32).reshape(shape)
    return out
