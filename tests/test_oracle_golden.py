"""CPU: the oracle restatement (oracle/tsdiff_oracle.py) against golden vectors produced by
the unchanged reference (oracle/gen_golden.py).  This is what PINS the oracle."""
import numpy as np
import pytest
import torch

from oracle import tsdiff_oracle as O
from tsdiff_amd import synth
from tests.util import assert_close, batch_inputs, load_golden

# fp32 re-association noise between two CPU evaluations of the same network
# (reference noise floor vs fp64 is 3e-6)
RTOL = 2e-5


def _sd(meta, key="seed"):
    return O.to_torch_state(synth.synth_state_dict(meta["cfg"], meta[key]))


@pytest.mark.parametrize("name", ["fwd_rxn0_b1_full", "fwd_rxn0_b4_sigma_full", "fwd_synth_b6_small"])
def test_forward_matches_reference(name):
    d, meta = load_golden(name)
    b = batch_inputs(d)
    trace = {}
    edge_inv, ei, el = O.forward(_sd(meta), meta["cfg"], b["atom_type"], b["r_feat"], b["p_feat"], b["pos"],
                                 b["bond_index"], b["bond_type"], b["num_nodes_per_graph"].numpy(), trace=trace)
    # integer / index work: bit exact
    assert np.array_equal(ei.numpy(), d["edge_index"])
    assert np.array_equal(trace["enc_edge_index"].numpy(), d["enc_edge_index"])
    assert np.array_equal(trace["enc_type_r"].numpy(), d["enc_type_r"])
    assert np.array_equal(trace["enc_type_p"].numpy(), d["enc_type_p"])
    assert np.array_equal(trace["out_type_r"].numpy(), d["out_type_r"])
    assert np.array_equal(trace["out_type_p"].numpy(), d["out_type_p"])
    assert_close(el.numpy(), d["edge_length"], 1e-6, "edge_length")
    if "enc_edge_attr" in d:
        assert_close(trace["enc_edge_attr"].numpy(), d["enc_edge_attr"], RTOL, "enc_edge_attr")
        assert_close(trace["out_edge_attr"].numpy(), d["out_edge_attr"], RTOL, "out_edge_attr")
    L = int(meta["cfg"]["encoder"]["num_convs"])
    assert_close(trace[f"h{L}"].numpy(), d["h_final"], RTOL, "h_final")
    assert_close(edge_inv.numpy(), d["edge_inv"], RTOL, "edge_inv")
    node_eq = O.eq_transform(edge_inv, b["pos"], ei, el)
    assert_close(node_eq.numpy(), d["node_eq"], RTOL, "node_eq")


def test_edge_sets_differ_in_sigma_case():
    """the fixture really exercises E_enc != E_out and C = 0 edges"""
    d, _ = load_golden("fwd_rxn0_b4_sigma_full")
    assert d["enc_edge_index"].shape[1] > d["edge_index"].shape[1]
    assert (d["edge_length"] > 10.0).any()


def test_ensemble_forward():
    d, meta = load_golden("ens_synth_b6_small")
    b = batch_inputs(d)
    sds = [O.to_torch_state(synth.synth_state_dict(meta["cfg"], s)) for s in meta["seeds"]]
    edge_inv, ei, el = O.ensemble_forward(sds, meta["cfg"], b["atom_type"], b["r_feat"], b["p_feat"], b["pos"],
                                          b["bond_index"], b["bond_type"], b["num_nodes_per_graph"].numpy())
    assert np.array_equal(ei.numpy(), d["edge_index"])
    assert_close(edge_inv.numpy(), d["edge_inv"], RTOL, "edge_inv")


@pytest.mark.parametrize("name", ["ld_rxn0_b1_full_50", "ld_synth_b3_small_ens2_20", "ddpm_synth_b3_small_12",
                                  "ld_guess_denoise_small", "ld_guess_noise_denoise_small"])
def test_sampler_trajectory(name):
    d, meta = load_golden(name)
    b = batch_inputs(d)
    sds = [O.to_torch_state(synth.synth_state_dict(meta["cfg"], s)) for s in meta["seeds"]]
    pos, traj = O.sample(sds, meta["cfg"], b["atom_type"], b["r_feat"], b["p_feat"],
                         torch.from_numpy(d["pos_init"]), b["bond_index"], b["bond_type"], b["batch"],
                         b["num_nodes_per_graph"].numpy(), torch.from_numpy(d["noises"]),
                         n_steps=meta["n_steps"], step_lr=meta["step_lr"], clip=meta["clip"],
                         sampling_type=meta["sampling_type"],
                         denoise_from_time_t=meta.get("denoise_from_time_t"),
                         noise_from_time_t=meta.get("noise_from_time_t"),
                         init_noise=torch.from_numpy(d["init_noise"]) if "init_noise" in d else None)
    # trajectories amplify rounding differences slightly; positions are O(10) Angstrom
    assert_close(torch.stack(traj).numpy(), d["traj"], 5e-5, "traj")
    assert_close(pos.numpy(), d["pos_final"], 5e-5, "pos_final")


@pytest.mark.parametrize("name", ["loss_synth_b4_small", "loss_rxn0_b2_full"])
def test_get_loss(name):
    d, meta = load_golden(name)
    b = batch_inputs(d)
    loss = O.get_loss(_sd(meta), meta["cfg"], b["atom_type"], b["r_feat"], b["p_feat"], b["pos"],
                      b["bond_index"], b["bond_type"], b["batch"], b["num_nodes_per_graph"].numpy(),
                      torch.from_numpy(d["time_step"]), torch.from_numpy(d["pos_noise"]))
    assert_close(loss.numpy(), d["loss"], 5e-5, "loss")


def test_get_loss_gradients_elementwise():
    """pins the oracle's autograd: EVERY parameter gradient of loss.mean() against the unchanged reference's,
    element by element (the GPU parity tests use the oracle's autograd as the checker for the full-size model)"""
    d, meta = load_golden("grads_synth_b4_small")
    b = batch_inputs(d)
    sd = _sd(meta)
    for v in sd.values():
        v.requires_grad_(True)
    loss = O.get_loss(sd, meta["cfg"], b["atom_type"], b["r_feat"], b["p_feat"], b["pos"], b["bond_index"],
                      b["bond_type"], b["batch"], b["num_nodes_per_graph"].numpy(), torch.from_numpy(d["time_step"]),
                      torch.from_numpy(d["pos_noise"]))
    assert_close(loss.detach().numpy(), d["loss"], 5e-5, "loss")
    loss.mean().backward()
    keys = [k[5:] for k in d if k.startswith("grad.")]
    assert len(keys) == len(meta["grad_norms"]) >= 30
    for k in keys:
        assert sd[k].grad is not None, k
        assert_close(sd[k].grad.numpy(), d["grad." + k], 5e-5, "d loss / d " + k)


def test_schedule():
    d, meta = load_golden("schedule_full")
    betas, alphas = O.beta_schedule(meta["cfg"])
    assert np.array_equal(betas.numpy(), d["betas"])
    assert np.array_equal(alphas.numpy(), d["alphas"])
    sig = O.sigmas_from_alphas(alphas)
    assert abs(float(sig[-1]) - 12.1685) < 1e-3


def test_cfconv_aggregate_symmetric_rows():
    """aggregation over destination == aggregation over the row segment when the edge set,
    and W, are symmetric -- the identity the HIP kernel relies on."""
    d, meta = load_golden("fwd_synth_b6_small")
    b = batch_inputs(d)
    trace = {}
    O.forward(_sd(meta), meta["cfg"], b["atom_type"], b["r_feat"], b["p_feat"], b["pos"],
              b["bond_index"], b["bond_type"], b["num_nodes_per_graph"].numpy(), trace=trace)
    ei, W, x1 = trace["enc_edge_index"], trace["W0"], trace["x1_0"]
    N = x1.shape[0]
    by_row = torch.zeros(N, W.shape[1]).index_add_(0, ei[0], x1[ei[1]] * W)
    assert_close(by_row.numpy(), trace["agg0"].numpy(), 1e-6, "row-vs-col aggregation")


def test_smooth_conv_forward():
    """encoder.smooth_conv = True (cosine cutoff, reference schnet.py:92-96)"""
    d, meta = load_golden("fwd_synth_b6_small_smooth")
    assert meta["cfg"]["encoder"]["smooth_conv"] is True
    b = batch_inputs(d)
    trace = {}
    edge_inv, ei, el = O.forward(_sd(meta), meta["cfg"], b["atom_type"], b["r_feat"], b["p_feat"], b["pos"],
                                 b["bond_index"], b["bond_type"], b["num_nodes_per_graph"].numpy(), trace=trace)
    assert np.array_equal(ei.numpy(), d["edge_index"])
    assert_close(trace["h2"].numpy(), d["h_final"], RTOL, "h_final")
    assert_close(edge_inv.numpy(), d["edge_inv"], RTOL, "edge_inv")
