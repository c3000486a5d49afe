"""CPU: the dual-encoder oracle (oracle/dualenc_oracle.py) against goldens produced by the unchanged reference
`DualEncoderEpsNetwork` (oracle/gen_golden.py section H).  This PINS that oracle.  Also checks the host
mirror's state_dict key set against the reference's."""
import numpy as np
import pytest
import torch

from oracle import dualenc_oracle as D
from oracle import tsdiff_oracle as O
from tsdiff_amd import synth
from tests.util import assert_close, batch_inputs, load_golden

RTOL = 2e-5


def _state(d):
    return {k[3:]: torch.from_numpy(v) for k, v in d.items() if k.startswith("sd.")}


def _check_fwd(d, tag, out):
    inv_g, inv_l, ei, et, el, local = out
    assert np.array_equal(ei.numpy(), d[f"{tag}.edge_index"]), tag
    assert np.array_equal(et.numpy(), d[f"{tag}.edge_type"]), tag
    assert np.array_equal(local.numpy(), d[f"{tag}.local_edge_mask"]), tag
    assert_close(el.numpy(), d[f"{tag}.edge_length"], 1e-6, tag + " edge_length")
    assert_close(inv_g.numpy(), d[f"{tag}.edge_inv_global"], RTOL, tag + " edge_inv_global")
    assert_close(inv_l.numpy(), d[f"{tag}.edge_inv_local"], RTOL, tag + " edge_inv_local")


@pytest.mark.parametrize("name", ["dual_small", "dual_small_ts"])
def test_forward_variants(name):
    d, meta = load_golden(name)
    b = batch_inputs(d)
    sd, cfg = _state(d), meta["cfg"]
    nn_ = b["num_nodes_per_graph"].numpy()
    args = (b["atom_type"],)
    with torch.no_grad():
        _check_fwd(d, "fwd", D.forward(sd, cfg, b["atom_type"], b["pos"], b["bond_index"], b["bond_type"], nn_))
        far = b["pos"] * 4.0
        _check_fwd(d, "far", D.forward(sd, cfg, b["atom_type"], far, b["bond_index"], b["bond_type"], nn_))
        _check_fwd(d, "noorder", D.forward(sd, cfg, b["atom_type"], far, b["bond_index"], b["bond_type"], nn_,
                                           extend_order=False))
        _check_fwd(d, "noradius", D.forward(sd, cfg, b["atom_type"], b["pos"], b["bond_index"], b["bond_type"], nn_,
                                            extend_radius=False))
    # the fixtures exercise what they claim to
    assert (d["far.edge_length"] > 10.0).any() and (d["far.edge_type"] == 0).any()
    assert d["noorder.edge_index"].shape[1] < d["far.edge_index"].shape[1] or (d["noorder.edge_type"] < 484).all()
    assert (d["noradius.edge_type"] > 0).all()
    w = sd["encoder_global.node_emb.weight"]
    used = torch.unique(b["atom_type"])
    assert (w[used].norm(dim=1) > 10).any() and (w[used].norm(dim=1) < 10).any()  # max_norm binds for some rows


@pytest.mark.parametrize("name", ["dual_small", "dual_small_ts"])
def test_loss_and_gradients(name):
    d, meta = load_golden(name)
    b = batch_inputs(d)
    sd, cfg = _state(d), meta["cfg"]
    for v in sd.values():
        if v.is_floating_point():
            v.requires_grad_(True)
    loss, lg, ll = D.get_loss(sd, cfg, b["atom_type"], b["pos"], b["bond_index"], b["bond_type"], b["batch"],
                              b["num_nodes_per_graph"].numpy(), torch.from_numpy(d["loss.time_step"]),
                              torch.from_numpy(d["loss.pos_noise"]))
    assert_close(loss.detach().numpy(), d["loss.loss"], 1e-4, "loss")
    assert_close(lg.detach().numpy(), d["loss.loss_global"], 1e-4, "loss_global")
    assert_close(ll.detach().numpy(), d["loss.loss_local"], 1e-4, "loss_local")
    loss.mean().backward()
    n = 0
    for k, v in sd.items():
        if "grad." + k in d:
            assert_close(v.grad.numpy(), d["grad." + k], 2e-4, "grad " + k)
            n += 1
    assert n >= 3
    for k, ref in meta["grad_norms"].items():
        got = float(sd[k].grad.norm())
        assert abs(got - ref) <= 2e-4 * max(ref, 1e-12), (k, got, ref)
    assert len(meta["grad_norms"]) >= 30


@pytest.mark.parametrize("st,kw", [
    ("ld", dict(sampling_type="ld", step_lr=1e-6)),
    ("ddpm_noisy", dict(clip_local=3.0)),
    ("ddpm_det", dict(sampling_type="ddpm_det", global_start_sigma=0.5, clip_pos=40.0)),
    ("generalized", dict(sampling_type="generalized", eta=0.7, w_global=0.35)),
])
def test_sampler_trajectories(st, kw):
    d, meta = load_golden("dual_small")
    b = batch_inputs(d)
    sd, cfg = _state(d), meta["cfg"]
    noise = torch.from_numpy(d[f"samp.{st}.noise"])
    with torch.no_grad():
        pos, traj = D.sample(sd, cfg, b["atom_type"], torch.from_numpy(d["samp.pos_init"]), b["bond_index"],
                             b["bond_type"], b["batch"], b["num_nodes_per_graph"].numpy(), noise, noise.shape[0], **kw)
    assert_close(torch.stack(traj).numpy(), d[f"samp.{st}.traj"], 1e-4, st)


def test_qm9_default_config_forward():
    d, meta = load_golden("dual_qm9_fwd")
    b = batch_inputs(d)
    from tsdiff_amd.epsnet import get_model
    model = get_model(meta["cfg"])
    shapes = [(k, tuple(dict(model.state_dict())[k].shape)) for k in meta["names"]]
    sd = {k: torch.from_numpy(v) for k, v in synth.hash_state_dict(shapes, meta["seed"]).items()}
    for k in model.state_dict():
        if k.endswith(".eps"):
            sd[k] = torch.zeros(1)
    with torch.no_grad():
        inv_g, inv_l, ei, et, el, local = D.forward(sd, meta["cfg"], b["atom_type"], torch.from_numpy(d["pos"]),
                                                    b["bond_index"], b["bond_type"], b["num_nodes_per_graph"].numpy())
    assert np.array_equal(et.numpy(), d["edge_type"])
    assert_close(inv_g.numpy(), d["edge_inv_global"], RTOL, "edge_inv_global")
    assert_close(inv_l.numpy(), d["edge_inv_local"], RTOL, "edge_inv_local")


@pytest.mark.parametrize("name", ["dual_small", "dual_small_ts"])
def test_host_mirror_state_dict_keys(name):
    _, meta = load_golden(name)
    from tsdiff_amd.epsnet import get_model
    model = get_model(meta["cfg"])
    assert sorted(model.state_dict().keys()) == meta["state_dict_keys"]
