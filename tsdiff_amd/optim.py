"""The optimizer half of the training step (reference train.py:103-104,144-145, utils/common.py:58-90) on the FLAT
parameter and gradient vectors of the fused step.

The fused training step (tsdiff_amd/train_ops.py) produces every parameter gradient as a view of ONE flat fp32
buffer.  `flatten_parameters(model)` makes the parameters views of one flat buffer too (same order), after which

    optimizer = get_optimizer(config.train.optimizer, model)      # utils.common.get_optimizer's signature
    ...
    orig_grad_norm = clip_grad_norm_(model.parameters(), config.train.max_grad_norm)
    optimizer.step()

run as three + one launches over the flat vectors (tsd_grad_norm_clip, tsd_adam_step) instead of torch's ~25
multi-tensor launches over 80 tensors.  Same update rule and state layout as torch.optim.Adam: `state_dict()` /
`load_state_dict()` exchange checkpoints with it (per-parameter `step`, `exp_avg`, `exp_avg_sq`).  Anything that is
not laid out flat (other models, gradients from the op-by-op path) takes torch's own implementation."""
import ctypes as C

import torch
from torch.optim.adam import adam as _torch_adam

from . import _lib
from ._lib import check, ptr, stream_ptr


def flatten_parameters(model):
    """Re-home the trainable parameters of a CondenseEncoderEpsNetwork in ONE flat fp32 buffer, in the order of the
    flat parameter vector (engine.raw_param_names); the Parameter objects stay (state_dict, optimizers and hooks
    keep working), only their storage moves.  Returns the flat tensor.  Idempotent."""
    params = model.raw_params()
    flat = getattr(model, "_flat_param", None)
    if flat is not None and _views_of(params, flat):
        return flat
    dev, dt = params[0].device, params[0].dtype
    flat = torch.empty(sum(p.numel() for p in params), dtype=dt, device=dev)
    o = 0
    with torch.no_grad():
        for p in params:
            n = p.numel()
            flat[o:o + n].copy_(p.detach().reshape(-1))
            p.data = flat[o:o + n].view(p.shape)
            o += n
    model._flat_param = flat
    return flat


def _views_of(tensors, flat):
    """True when `tensors`, in order, tile `flat` exactly (consecutive, contiguous views of its storage)."""
    if flat is None or not flat.is_contiguous():
        return False
    o = flat.storage_offset()
    sp = flat.untyped_storage().data_ptr()
    for t in tensors:
        if t is None or t.untyped_storage().data_ptr() != sp or t.storage_offset() != o or not t.is_contiguous() \
                or t.dtype != flat.dtype:
            return False
        o += t.numel()
    return o == flat.storage_offset() + flat.numel()


def _by_offset(tensors):
    """`tensors` ordered by their position in memory when they share one storage (model.parameters() walks the
    module tree, the flat vector is in engine.raw_param_names order), else unchanged"""
    if tensors and all(t is not None for t in tensors) and \
            len({t.untyped_storage().data_ptr() for t in tensors}) == 1:
        return sorted(tensors, key=lambda t: t.storage_offset())
    return list(tensors)


def _flat_base(tensors):
    """the flat tensor that `tensors` tile in order (one storage, consecutive, contiguous), or None"""
    if not tensors or any(t is None for t in tensors):
        return None
    t0 = tensors[0]
    n = sum(t.numel() for t in tensors)
    st = t0.untyped_storage()
    if (t0.storage_offset() + n) * t0.element_size() > st.nbytes():
        return None
    flat = torch.empty(0, dtype=t0.dtype, device=t0.device).set_(st, t0.storage_offset(), (n,), (1,))
    return flat if _views_of(tensors, flat) else None


class _Layout:
    """A parameter list that tiles one flat buffer, ordered by address, with the byte offset of every member: the
    per-step question "do the gradients tile a flat buffer the same way" is then one data_ptr() per gradient."""

    def __init__(self, params):
        self.params = _by_offset(params)
        self.flat = _flat_base(self.params)
        self.ids = tuple(id(p) for p in params)
        self.n = sum(p.numel() for p in self.params)
        self.offsets, o = [], 0
        for p in self.params:
            self.offsets.append(o * 4)
            o += p.numel()
        self.first = self.params[0].data_ptr() if self.params else 0

    def params_flat(self):
        """the flat parameter buffer, or None (never flat, or the parameters were re-homed since)"""
        if self.flat is None or self.params[0].data_ptr() != self.first or \
                self.params[-1].data_ptr() != self.first + self.offsets[-1]:
            return None
        return self.flat

    def grads_flat(self):
        """the flat fp32 gradient buffer whose views, in address order of the parameters, are their .grad -- or None"""
        g0 = self.params[0].grad
        if g0 is None or g0.dtype != torch.float32 or not g0.is_cuda:
            return None
        base = g0.data_ptr()
        for p, off in zip(self.params, self.offsets):
            g = p.grad
            if g is None or g.data_ptr() != base + off or g.numel() != p.numel():
                return None
        st = g0.untyped_storage()
        if (g0.storage_offset() + self.n) * 4 > st.nbytes():
            return None
        return torch.empty(0, dtype=torch.float32, device=g0.device).set_(st, g0.storage_offset(), (self.n,), (1,))


_clip_layouts = {}


def clip_grad_norm_(parameters, max_norm, norm_type=2.0, error_if_nonfinite=False, foreach=None):
    """torch.nn.utils.clip_grad_norm_ (reference train.py:144).  Gradients that are views of one flat buffer (the
    fused training step's) are normed and scaled by three launches over that buffer, without a host sync; the total
    norm comes back as a 0-dim tensor, as torch's does."""
    if isinstance(parameters, torch.Tensor):
        parameters = [parameters]
    params = [p for p in parameters if p.grad is not None]
    flat = None
    # (max_norm <= 0: torch zeroes / flips the gradients; not worth a kernel variant -- torch's own path)
    if params and float(norm_type) == 2.0 and not error_if_nonfinite and float(max_norm) > 0.0:
        key = tuple(id(p) for p in params)
        lay = _clip_layouts.get(key)
        if lay is None:
            _clip_layouts.clear()  # one model at a time: the cache must not keep dead parameters' ids alive
            lay = _clip_layouts[key] = _Layout(params)
        flat = lay.grads_flat()
    if flat is None:
        return torch.nn.utils.clip_grad_norm_(params, max_norm, norm_type=norm_type,
                                              error_if_nonfinite=error_if_nonfinite, foreach=foreach)
    lib = _lib.load()
    scratch = torch.empty(1025, dtype=torch.float32, device=flat.device)
    check(lib.tsd_grad_norm_clip(flat.numel(), ptr(flat), float(max_norm), ptr(scratch), ptr(scratch[1024:]),
                                 stream_ptr()))
    return scratch[1024]


class Adam(torch.optim.Optimizer):
    """torch.optim.Adam's update rule (no amsgrad / maximize / capturable) with a one-launch path for parameters and
    gradients that are flat (see the module docstring); the state layout is torch.optim.Adam's."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0.0 or eps < 0.0 or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or weight_decay < 0.0:
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._flat_state = {}  # group index -> (flat param it was built for, exp_avg, exp_avg_sq, shared step)
        self._layouts = {}     # group index -> _Layout of the group's parameters that receive gradients

    def _ensure_flat_state(self, gi, params, flat_p):
        """flat exp_avg / exp_avg_sq whose per-parameter views ARE the entries of self.state (so state_dict() is
        torch.optim.Adam's) and one step counter shared by the group's parameters; existing per-parameter state (a
        loaded checkpoint) is copied in.  None when the parameters disagree about the step count."""
        fs = self._flat_state.get(gi)
        if fs is not None and fs[0] is flat_p and all(
                self.state[p].get("exp_avg") is not None and self.state[p]["exp_avg"]._base is fs[1] for p in params):
            return fs
        steps = {float(self.state[p]["step"]) for p in params if "step" in self.state[p]}
        if len(steps) > 1 or (steps and any("step" not in self.state[p] for p in params)):
            return None
        t = torch.tensor(steps.pop() if steps else 0.0, dtype=torch.float32)
        m = torch.zeros_like(flat_p)
        v = torch.zeros_like(flat_p)
        o = 0
        for p in params:
            n = p.numel()
            st = self.state[p]
            mv, vv = m[o:o + n].view(p.shape), v[o:o + n].view(p.shape)
            if "exp_avg" in st:
                mv.copy_(st["exp_avg"])
                vv.copy_(st["exp_avg_sq"])
            st["exp_avg"], st["exp_avg_sq"], st["step"] = mv, vv, t
            o += n
        fs = (flat_p, m, v, t)
        self._flat_state[gi] = fs
        return fs

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            params = [p for p in group["params"] if p.grad is not None]
            if not params:
                continue
            b1, b2 = group["betas"]
            lay = self._layouts.get(gi)
            if lay is None or lay.ids != tuple(id(p) for p in params):
                lay = self._layouts[gi] = _Layout(params)
            flat_p = lay.params_flat()
            flat_g = lay.grads_flat() if flat_p is not None and flat_p.is_cuda and flat_p.dtype == torch.float32 else None
            if flat_g is not None:
                fs = self._ensure_flat_state(gi, lay.params, flat_p)
                if fs is not None:
                    _, m, v, t = fs
                    t += 1  # one CPU scalar shared by the group's parameters
                    check(_lib.load().tsd_adam_step(flat_p.numel(), ptr(flat_p), ptr(flat_g), ptr(m), ptr(v),
                                                    float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                                    float(group["weight_decay"]), int(t), stream_ptr()))
                    # the update wrote the parameters behind autograd's back: bump the version counters that caches
                    # (the packed inference weights) key on
                    for p in params:
                        torch.autograd.graph.increment_version(p)
                    continue
            self._torch_step(group, params)
        return loss

    def _torch_step(self, group, params):
        grads, ms, vs, steps = [], [], [], []
        for p in params:
            st = self.state[p]
            if "exp_avg" not in st:
                st["step"] = torch.tensor(0.0, dtype=torch.float32)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            grads.append(p.grad)
            ms.append(st["exp_avg"])
            vs.append(st["exp_avg_sq"])
            steps.append(st["step"])
        if len({id(t) for t in steps}) != len(steps):  # the flat path's shared counter: torch bumps one per parameter
            steps = []
            for p in params:
                self.state[p]["step"] = self.state[p]["step"].clone()
                steps.append(self.state[p]["step"])
            self._flat_state = {}
        b1, b2 = group["betas"]
        _torch_adam(params, grads, ms, vs, [], steps, amsgrad=False, beta1=b1, beta2=b2, lr=group["lr"],
                              weight_decay=group["weight_decay"], eps=group["eps"], maximize=False, foreach=None,
                              capturable=False, differentiable=False, fused=None, grad_scale=None, found_inf=None,
                              has_complex=False)

    def state_dict(self):
        """torch.optim.Adam's layout.  The flat path keeps ONE step counter per group, shared by the per-parameter
        state entries; a checkpoint must not carry that aliasing (torch.optim.Adam.load_state_dict keeps `step`
        tensors as they are, so its step() would bump the shared counter once per parameter): every entry gets its
        own copy of the counter here."""
        sd = super().state_dict()
        sd["state"] = {k: ({**st, "step": st["step"].clone()} if torch.is_tensor(st.get("step")) else dict(st))
                       for k, st in sd["state"].items()}
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._flat_state = {}  # the loaded per-parameter tensors are re-homed in flat buffers at the next step
        for st in self.state.values():  # torch keeps `step` on the CPU for the non-capturable path
            if "step" in st and torch.is_tensor(st["step"]):
                st["step"] = st["step"].detach().to("cpu", torch.float32)


def get_optimizer(cfg, model):
    """utils/common.py:58-70 with the flat fast path: `type: adam` only, as the reference."""
    if cfg.type != "adam":
        raise NotImplementedError("Optimizer not supported: %s" % cfg.type)
    if hasattr(model, "raw_params"):
        flatten_parameters(model)
    return Adam(model.parameters(), lr=cfg.lr, weight_decay=cfg.weight_decay, betas=(cfg.beta1, cfg.beta2))
