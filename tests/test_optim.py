"""CPU: tsdiff_amd.optim's fallback path (parameters / gradients that are not flat) is torch.optim.Adam and
torch.nn.utils.clip_grad_norm_ themselves; the flat-layout detection; get_optimizer mirrors utils/common.py:58-70."""
from types import SimpleNamespace

import pytest
import torch

from tsdiff_amd import optim


def _net(seed):
    torch.manual_seed(seed)
    return torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Tanh(), torch.nn.Linear(7, 3))


def test_fallback_equals_torch_adam_and_clip():
    a, b = _net(0), _net(0)
    oa = torch.optim.Adam(a.parameters(), lr=1e-2, betas=(0.95, 0.999), weight_decay=1e-2)
    ob = optim.Adam(b.parameters(), lr=1e-2, betas=(0.95, 0.999), weight_decay=1e-2)
    x = torch.randn(11, 5)
    for _ in range(4):
        for m, o, clip in ((a, oa, torch.nn.utils.clip_grad_norm_), (b, ob, optim.clip_grad_norm_)):
            o.zero_grad()
            m(x).square().mean().backward()
            n = clip(m.parameters(), 0.3)
            o.step()
        assert torch.is_tensor(n)
    for p, q in zip(a.parameters(), b.parameters()):
        assert torch.equal(p, q)
    sa, sb = oa.state_dict(), ob.state_dict()
    assert sa["param_groups"][0]["lr"] == sb["param_groups"][0]["lr"] and set(sa["state"]) == set(sb["state"])
    for k in sa["state"]:
        assert torch.equal(sa["state"][k]["exp_avg_sq"], sb["state"][k]["exp_avg_sq"])
        assert float(sa["state"][k]["step"]) == float(sb["state"][k]["step"]) == 4.0
    oa2 = torch.optim.Adam(a.parameters(), lr=1e-2, betas=(0.95, 0.999), weight_decay=1e-2)
    oa2.load_state_dict(sb)  # our checkpoints load into torch's optimizer


def test_flat_layout_detection():
    flat = torch.arange(24, dtype=torch.float32)
    views = [flat[0:6].view(2, 3), flat[6:10], flat[10:24].view(7, 2)]
    assert optim._flat_base(views).data_ptr() == flat.data_ptr() and optim._flat_base(views).numel() == 24
    assert optim._flat_base([views[0], views[2]]) is None          # a gap
    assert optim._flat_base([views[1], views[0]]) is None          # out of order
    assert [v.data_ptr() for v in optim._by_offset([views[2], views[0], views[1]])] == [v.data_ptr() for v in views]
    assert optim._flat_base([torch.zeros(3), torch.zeros(3)]) is None
    lay = optim._Layout([views[2], views[0], views[1]])
    assert lay.params_flat().data_ptr() == flat.data_ptr() and lay.offsets == [0, 24, 40] and lay.n == 24
    sub = optim._flat_base(views[1:])                               # a suffix of the buffer tiles a slice of it
    assert sub.numel() == 18 and sub.data_ptr() == flat[6:].data_ptr()
    assert optim._flat_base([views[0].t()]) is None                 # not contiguous


def test_get_optimizer_mirrors_the_reference():
    net = _net(1)
    cfg = SimpleNamespace(type="adam", lr=5e-4, weight_decay=0.0, beta1=0.95, beta2=0.999)
    o = optim.get_optimizer(cfg, net)
    g = o.param_groups[0]
    assert g["lr"] == 5e-4 and g["betas"] == (0.95, 0.999) and g["weight_decay"] == 0.0 and g["eps"] == 1e-8
    torch.optim.lr_scheduler.ReduceLROnPlateau(o, factor=0.6, patience=10, min_lr=1e-6)  # utils/common.py:73-80
    with pytest.raises(NotImplementedError):
        optim.get_optimizer(SimpleNamespace(type="sgd"), net)


def test_flatten_parameters_keeps_the_module_surface():
    """optim.flatten_parameters on the real network (CPU: no kernel runs): the parameters become views of one flat
    buffer in the order of the flat parameter vector, state_dict / load_state_dict / named_parameters are unchanged,
    and an optimizer step on synthetic gradients (CPU tensors: torch's own update rule) equals torch.optim.Adam on an
    unflattened copy"""
    from tsdiff_amd import engine, synth
    from tsdiff_amd.epsnet import get_model
    from tsdiff_amd.utils import AttrDict
    cfg = synth.small_model_config(64, 2)
    sd = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(cfg, 1).items()}
    a, b = get_model(AttrDict(cfg)), get_model(AttrDict(cfg))
    a.load_state_dict(sd, strict=False), b.load_state_dict(sd, strict=False)
    keys = list(b.state_dict().keys())
    flat = optim.flatten_parameters(b)
    assert optim.flatten_parameters(b) is flat
    names = engine.raw_param_names(b._cfg.num_convs)
    P = dict(b.named_parameters())
    assert flat.numel() == sum(P[n].numel() for n in names)
    o = 0
    for n in names:  # consecutive views, raw order
        assert P[n].data_ptr() == flat.data_ptr() + 4 * o and P[n].is_contiguous()
        o += P[n].numel()
    assert list(b.state_dict().keys()) == keys
    for k, v in a.state_dict().items():
        assert torch.equal(v, b.state_dict()[k])
    b.load_state_dict(a.state_dict())          # copies INTO the views
    assert P[names[0]].data_ptr() == flat.data_ptr()
    oa = torch.optim.Adam(a.parameters(), lr=1e-3, betas=(0.95, 0.999))
    ob = optim.get_optimizer(SimpleNamespace(type="adam", lr=1e-3, weight_decay=0.0, beta1=0.95, beta2=0.999), b)
    gen = torch.Generator().manual_seed(0)
    for _ in range(2):
        grads = {n: torch.randn(p.shape, generator=gen) for n, p in a.named_parameters() if p.requires_grad}
        for m in (a, b):
            for n, p in m.named_parameters():
                p.grad = grads[n].clone() if n in grads else None
        na = torch.nn.utils.clip_grad_norm_(a.parameters(), 1.0)
        nb = optim.clip_grad_norm_(b.parameters(), 1.0)
        assert torch.allclose(na, nb, rtol=1e-6)
        oa.step(), ob.step()
    for (n, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        assert torch.allclose(p, q, rtol=1e-6, atol=1e-8), n
    assert P[names[-1]].data_ptr() + 4 * P[names[-1]].numel() == flat.data_ptr() + 4 * flat.numel()


def test_state_dict_does_not_export_the_shared_step_counter():
    """the flat path keeps ONE step tensor per group inside self.state; state_dict() must hand every parameter its
    own copy, or torch.optim.Adam (which keeps loaded `step` tensors as they are) bumps it once per parameter.
    CPU: the flat state is built by hand on a flat CPU buffer (no kernel runs)."""
    import copy
    flat = torch.zeros(5 * 7 + 7 + 7 * 3 + 3)
    net = _net(3)
    o = 0
    for p in net.parameters():
        n = p.numel()
        flat[o:o + n] = p.detach().reshape(-1)
        p.data = flat[o:o + n].view(p.shape)
        o += n
    params = list(net.parameters())
    opt = optim.Adam(params, lr=1e-2)
    fs = opt._ensure_flat_state(0, params, flat)
    assert fs is not None
    fs[3].add_(1)  # what a flat step does to the shared counter
    assert len({id(opt.state[p]["step"]) for p in params}) == 1  # shared inside the optimizer ...
    sd = opt.state_dict()
    assert len({id(st["step"]) for st in sd["state"].values()}) == len(params)  # ... not in the checkpoint
    assert all(float(st["step"]) == 1.0 for st in sd["state"].values())
    assert len({id(opt.state[p]["step"]) for p in params}) == 1  # and state_dict() left the live state alone
    for loader in (copy.deepcopy, lambda x: x):
        net2 = _net(3)
        o2 = torch.optim.Adam(net2.parameters(), lr=1e-2)
        o2.load_state_dict(loader(sd))
        net2(torch.randn(4, 5)).square().mean().backward()
        o2.step()
        assert all(float(st["step"]) == 2.0 for st in o2.state.values())


def test_clip_with_nonpositive_max_norm_is_torchs():
    a, b = _net(2), _net(2)
    x = torch.randn(6, 5)
    for m in (a, b):
        m(x).square().mean().backward()
    torch.nn.utils.clip_grad_norm_(a.parameters(), 0.0)
    optim.clip_grad_norm_(b.parameters(), 0.0)
    for p, q in zip(a.parameters(), b.parameters()):
        assert torch.equal(p.grad, q.grad) and float(p.grad.abs().sum()) == 0.0
