"""SURVEY §8(f) N2 — what surrounds the hot path in the reference's train step, each as ONE HIP launch.

Drop-ins for the reference's classes (same constructor arguments and call signatures):

  FocalLoss   <- src/losses.py:53-66      (registered as "focal_loss" in src/argus_models.py:22-25)
  FusedAdamW  <- torch.optim.AdamW        (argus builds it from ("AdamW", {...}), configs/ball_action/*.py:51)
  FusedSGD    <- torch.optim.SGD          (("SGD", {momentum 0.9, nesterov}), configs/ball_action/ball_finetune_long_004.py:51-55)
  ModelEma    <- src/ema.py:13-55         (created in scripts/ball_action/train.py:80-82, updated every step,
                                           src/argus_models.py:65-66)

torch runs these as ~15 (loss fwd+bwd) + 9 (fused AdamW) + ~1500 (ModelEma: three ops per state_dict entry)
launches per step; here: 2 + 1 + ~7.  As everywhere in this package there is no CPU fallback: CPU tensors raise
unless the test-suite's kernel simulator has been injected (`mds.train.LIB = ...`).
"""
from __future__ import annotations

import ctypes as C
from copy import deepcopy

import torch
from torch import nn

from . import cabi

LIB = None            # tests inject the kernel simulator here; product: cabi.load()
REDUCTIONS = {"none": cabi.MDS_REDUCE_NONE, "mean": cabi.MDS_REDUCE_MEAN, "sum": cabi.MDS_REDUCE_SUM}
CHUNK = cabi.MDS_OPT_CHUNK


def _lib(t: torch.Tensor):
    if LIB is not None:
        return LIB
    if not t.is_cuda:
        raise cabi.MdsError("mds.train runs on MI355X only: move the tensors to cuda")
    return cabi.load()


def _stream(t: torch.Tensor):
    return torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else 0


# ------------------------------------------------------------------------------------------------ focal loss
class _Focal(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inputs, targets, alpha, gamma, reduction):
        x = inputs.detach().float().contiguous()
        t = targets.detach().float().contiguous()
        assert x.shape == t.shape, "focal loss: inputs and targets must have the same shape"
        n = x.numel()
        red = REDUCTIONS[reduction]
        loss = torch.zeros(n if red == cabi.MDS_REDUCE_NONE else 1, dtype=torch.float32, device=x.device)
        dx = torch.empty_like(x)
        lib = _lib(x)
        with torch.cuda.device(x.device) if x.is_cuda else _null():
            args = cabi.make("mds_focal_args", n=n, x=x, t=t, alpha=float(alpha), gamma=float(gamma), reduction=red, loss=loss, dx=dx)
            lib.check(lib.fn["focal_fwd_bwd"](C.byref(args), _stream(x)), "focal_fwd_bwd")
        ctx.save_for_backward(dx)
        ctx.in_dtype = inputs.dtype
        return loss.view(x.shape) if red == cabi.MDS_REDUCE_NONE else loss.view(())

    @staticmethod
    def backward(ctx, gout):
        (dx,) = ctx.saved_tensors
        return (dx * gout).to(ctx.in_dtype), None, None, None, None


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def sigmoid_focal_loss(inputs, targets, alpha: float = -1.0, gamma: float = 2.0, reduction: str = "mean"):
    """src/losses.py:5-50 — value and gradient from one kernel."""
    return _Focal.apply(inputs, targets, alpha, gamma, reduction)


class FocalLoss(nn.Module):
    def __init__(self, alpha: float = -1.0, gamma: float = 2.0, reduction: str = "mean"):
        super().__init__()
        self.alpha, self.gamma, self.reduction = alpha, gamma, reduction

    def forward(self, inputs, targets):
        return sigmoid_focal_loss(inputs, targets, alpha=self.alpha, gamma=self.gamma, reduction=self.reduction)


# ------------------------------------------------------------------------------------------------ tables
def _chunk_table(numels, device):
    """[(tensor index, chunk index)] for every MDS_OPT_CHUNK-sized piece of every tensor, as a device int32 array"""
    rows = []
    for i, n in enumerate(numels):
        rows.extend((i, c) for c in range((n + CHUNK - 1) // CHUNK))
    return torch.tensor(rows, dtype=torch.int32).reshape(-1, 2).contiguous().to(device), len(rows)


def _tensor_table(entries, device):
    """entries: (ptr, goff, soff, n) -> device copy of an mds_opt_tensor array"""
    T = cabi.STRUCTS["mds_opt_tensor"]
    arr = (T * len(entries))()
    for k, (p, goff, soff, n) in enumerate(entries):
        arr[k].p, arr[k].goff, arr[k].soff, arr[k].n = p, goff, soff, n
    return torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)


# ------------------------------------------------------------------------------------------------ optimizers
class _FusedOptimizer(torch.optim.Optimizer):
    """What FusedAdamW and FusedSGD share: per parameter group, flat fp32 state buffers (exposed through `state[p][...]`
    as views, so `state_dict()` has torch's layout), the device tables of one launch over every tensor, and the
    GradScaler hand-shake.

    `_step_supports_amp_scaling`: `GradScaler.step` (src/argus_models.py:62) then calls `step()` directly with
    `self.grad_scale` / `self.found_inf` set to device tensors - no unscale pass over the gradients and no host
    synchronisation to decide whether to skip; the kernels divide the gradients by the scale on load and return
    at once when an inf / nan was found.

    One step counter per GROUP (torch keeps one per parameter): they differ only if a parameter of the group has no
    gradient on some steps, which the reference's training loop never does."""
    _step_supports_amp_scaling = True
    STATE_NAMES = ()

    def _group_state(self, gi, group):
        gs = self._g.get(gi)
        ps = list(group["params"])
        dev = ps[0].device
        if gs is not None and gs["device"] != dev:           # module.to(device) after the first step: move the state along
            old = gs
            gs = None
        else:
            old = None
        if gs is None:
            assert all(p.dtype == torch.float32 and p.device == dev and p.is_contiguous() for p in ps), \
                f"mds {type(self).__name__}: fp32 contiguous parameters on one device"
            soff, off = {}, 0
            for p in ps:
                soff[id(p)] = off
                off += (p.numel() + 3) // 4 * 4          # keep every tensor's state 16-byte aligned
            # the step counter lives on the DEVICE (two scalars used alternately: a launch reads one and writes the other), because
            # a step GradScaler skips on found_inf must not count (torch's fused optimizers rewind it on the device as well)
            gs = dict(soff=soff, cache=None, device=dev, k=0,
                      step_dev=(old["step_dev"].to(dev) if old else torch.zeros(2, device=dev)))
            if old:
                gs["k"] = old["k"]
            for name in self.STATE_NAMES:
                gs[name] = old[name].to(dev) if old else torch.zeros(off, device=dev)
            for p in ps:
                o = soff[id(p)]
                self.state[p] = {"step": torch.tensor(0.0),
                                 **{name: gs[name][o:o + p.numel()].view_as(p) for name in self.STATE_NAMES}}
            self._g[gi] = gs
        return gs

    def _tables(self, gs, active, grads):
        """device tables of this group's launch; rebuilt when the set of tensors, a parameter's or a gradient's address changes"""
        dev = active[0].device
        base = grads[0].untyped_storage().data_ptr()
        shared = all(g.untyped_storage().data_ptr() == base for g in grads)
        if not shared:
            base = 0
        sig = (shared, tuple(id(p) for p in active), tuple(p.data_ptr() for p in active), tuple(g.data_ptr() - base for g in grads))
        cache = gs["cache"]
        if cache is None or cache["sig"] != sig:
            entries = [(p.data_ptr(), (g.data_ptr() - base) // 4, gs["soff"][id(p)], p.numel()) for p, g in zip(active, grads)]
            chunks, nchunks = _chunk_table([p.numel() for p in active], dev)
            cache = gs["cache"] = dict(sig=sig, table=_tensor_table(entries, dev), chunks=chunks, nchunks=nchunks)
        cache["keep"] = grads            # the launch is asynchronous: the gradient buffer must outlive it
        return cache, base

    def _amp(self, dev):
        """(found_inf, grad_scale) device tensors set by GradScaler.step for this call, or None"""
        fi, sc = getattr(self, "found_inf", None), getattr(self, "grad_scale", None)
        fix = lambda t: None if t is None else t.to(device=dev, dtype=torch.float32).reshape(1)
        return fix(fi), fix(sc)

    def _launch(self, gs, group, active, cache, base, found_inf, grad_scale):
        raise NotImplementedError

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            active = [p for p in group["params"] if p.grad is not None]
            if not active:
                continue
            gs = self._group_state(gi, group)
            grads = [p.grad for p in active]
            assert all(g.dtype == torch.float32 and g.is_contiguous() for g in grads), f"mds {type(self).__name__}: fp32 contiguous gradients"
            cache, base = self._tables(gs, active, grads)
            dev = active[0].device
            found_inf, grad_scale = self._amp(dev)
            with torch.cuda.device(dev) if dev.type == "cuda" else _null():
                self._launch(gs, group, active, cache, base, found_inf, grad_scale)
            gs["k"] ^= 1
            cache["amp"] = (found_inf, grad_scale)
            # the kernel wrote through raw pointers: tell autograd / version-based caches (mds.predict) that the parameters changed
            torch.autograd.graph.increment_version(active)
        return loss

    def state_dict(self):
        for gi, group in enumerate(self.param_groups):
            gs = self._g.get(gi)
            if gs is not None:
                step = float(gs["step_dev"][gs["k"]].item())
                for p in group["params"]:
                    self.state[p]["step"] = torch.tensor(step)
        return super().state_dict()

    def load_state_dict(self, sd):
        super().load_state_dict(sd)
        for gi, group in enumerate(self.param_groups):
            loaded = {id(p): self.state.get(p) for p in group["params"]}
            self._g.pop(gi, None)
            saved = {k: v for k, v in loaded.items() if v}
            for p in group["params"]:
                self.state.pop(p, None)
            gs = self._group_state(gi, group)
            for p in group["params"]:
                st = saved.get(id(p))
                if st:
                    for name in self.STATE_NAMES:
                        if st.get(name) is not None:
                            self.state[p][name].copy_(st[name])
                    if "step" in st:
                        gs["step_dev"][gs["k"]] = float(st["step"])


class FusedAdamW(_FusedOptimizer):
    """torch.optim.AdamW (decoupled weight decay, bias correction; no amsgrad / maximize) as one launch per param group.
    Gradients are addressed relative to one base pointer when they are views of one buffer - which is how
    mds.MultiDimStacker hands them over - so the device table is built once."""
    STATE_NAMES = ("exp_avg", "exp_avg_sq")

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False, *, maximize=False):
        if amsgrad or maximize:
            raise NotImplementedError("mds FusedAdamW: amsgrad / maximize are not implemented (no reference config uses them)")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._g = {}          # group index -> dict(exp_avg, exp_avg_sq, soff, step, cache, device)

    def _launch(self, gs, group, active, cache, base, found_inf, grad_scale):
        b1, b2 = group["betas"]
        k = gs["k"]
        lib = _lib(active[0])
        args = cabi.make("mds_adamw_args", table=cache["table"], chunks=cache["chunks"], nchunks=cache["nchunks"], gbase=base,
                         exp_avg=gs["exp_avg"], exp_avg_sq=gs["exp_avg_sq"], lr=float(group["lr"]), beta1=float(b1), beta2=float(b2),
                         eps=float(group["eps"]), weight_decay=float(group["weight_decay"]), bias1=0.0, bias2=0.0,
                         found_inf=found_inf, grad_scale=grad_scale, step_in=gs["step_dev"][k:k + 1], step_out=gs["step_dev"][1 - k:2 - k],
                         beta1_d=float(b1), beta2_d=float(b2))
        lib.check(lib.fn["multi_adamw"](C.byref(args), _stream(active[0])), "multi_adamw")


class FusedSGD(_FusedOptimizer):
    """torch.optim.SGD (momentum, dampening, weight_decay, nesterov; no maximize) as one launch per param group - the
    optimizer of the long-sequence fine-tune: ("SGD", {lr, momentum 0.9, nesterov True}),
    configs/ball_action/ball_finetune_long_004.py:51-55."""
    STATE_NAMES = ("momentum_buffer",)

    def __init__(self, params, lr=1e-3, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False, *, maximize=False):
        if maximize:
            raise NotImplementedError("mds FusedSGD: maximize is not implemented (no reference config uses it)")
        if nesterov and (momentum <= 0 or dampening != 0):
            raise ValueError("Nesterov momentum requires a momentum and zero dampening")
        super().__init__(params, dict(lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay, nesterov=nesterov))
        self._g = {}

    def _launch(self, gs, group, active, cache, base, found_inf, grad_scale):
        lib = _lib(active[0])
        args = cabi.make("mds_sgd_args", table=cache["table"], chunks=cache["chunks"], nchunks=cache["nchunks"], gbase=base,
                         momentum_buf=gs["momentum_buffer"], lr=float(group["lr"]), momentum=float(group["momentum"]),
                         dampening=float(group["dampening"]), weight_decay=float(group["weight_decay"]),
                         nesterov=int(bool(group["nesterov"])), first=0, found_inf=found_inf, grad_scale=grad_scale,
                         step_in=gs["step_dev"][gs["k"]:gs["k"] + 1], step_out=gs["step_dev"][1 - gs["k"]:2 - gs["k"]])
        lib.check(lib.fn["multi_sgd"](C.byref(args), _stream(active[0])), "multi_sgd")


# ------------------------------------------------------------------------------------------------ EMA
class ModelEma(nn.Module):
    """src/ema.py:13-55: a moving average of everything in the model's state_dict, iterated in order.
    Same-device float entries are updated by one multi-tensor launch; the integer entries (BatchNorm's
    num_batches_tracked) follow the reference's arithmetic (float blend, truncating cast) in a few batched ops."""

    def __init__(self, model, decay=0.9999, device=None):
        """The class contract of src/ema.py:37 (attribute names `ema`, `decay`, `device` are what callers read)."""
        super().__init__()
        shadow = deepcopy(model).eval()
        self.ema = shadow if device is None else shadow.to(device=device)
        self.decay, self.device = decay, device
        self._cache = None

    @torch.no_grad()
    def _blend_unfused(self, model, keep: float):
        """Off-device / non-fp32 path (EMA held on the CPU, `device='cpu'`): new = keep * ema + (1 - keep) * model for every
        state_dict entry, in state_dict order.  Float entries go through two `_foreach` calls; the integer counters keep the
        reference's arithmetic (src/ema.py:52: the blend promotes to float32 and `copy_` truncates)."""
        pairs = list(zip(self.ema.state_dict().values(), model.state_dict().values()))
        src = [m.detach().to(device=e.device) for e, m in pairs]
        fl = [(e, m.to(e.dtype)) for (e, _), m in zip(pairs, src) if e.is_floating_point()]
        if fl:
            if keep == 0.0:
                torch._foreach_copy_([e for e, _ in fl], [m for _, m in fl])
            else:
                part = torch._foreach_mul([m for _, m in fl], 1.0 - keep)       # three roundings, as `d * e + (1 - d) * m` has
                torch._foreach_mul_([e for e, _ in fl], keep)
                torch._foreach_add_([e for e, _ in fl], part)
        for (e, _), m in zip(pairs, src):
            if not e.is_floating_point():
                e.copy_(m if keep == 0.0 else keep * e + (1.0 - keep) * m)

    @torch.no_grad()
    def update(self, model):
        evs, mvs = list(self.ema.state_dict().values()), list(model.state_dict().values())
        fused_ok = self.device is None and all(e.device == m.device for e, m in zip(evs, mvs)) and (evs[0].is_cuda or LIB is not None)
        if not fused_ok:      # EMA kept on another device (src/ema.py `device='cpu'`): the reference's own loop
            return self._blend_unfused(model, float(self.decay))
        fl = [(e, m) for e, m in zip(evs, mvs) if e.dtype == torch.float32 and m.dtype == torch.float32 and e.is_contiguous() and m.is_contiguous()]
        rest = [(e, m) for e, m in zip(evs, mvs) if not (e.dtype == torch.float32 and m.dtype == torch.float32 and e.is_contiguous() and m.is_contiguous())]
        dev = evs[0].device
        sig = tuple((e.data_ptr(), m.data_ptr()) for e, m in fl)
        if self._cache is None or self._cache["sig"] != sig:
            entries = [(e.data_ptr(), m.data_ptr() // 4, 0, e.numel()) for e, m in fl]     # absolute source addresses (gbase = NULL)
            chunks, nchunks = _chunk_table([e.numel() for e, _ in fl], dev)
            self._cache = dict(sig=sig, table=_tensor_table(entries, dev) if fl else None, chunks=chunks, nchunks=nchunks)
        c = self._cache
        if c["nchunks"]:       # (a state_dict without fp32 entries has nothing to launch)
            lib = _lib(evs[0])
            args = cabi.make("mds_ema_args", table=c["table"], chunks=c["chunks"], nchunks=c["nchunks"], gbase=None, decay=float(self.decay))
            with torch.cuda.device(dev) if dev.type == "cuda" else _null():
                lib.check(lib.fn["multi_ema"](C.byref(args), _stream(evs[0])), "multi_ema")
            torch.autograd.graph.increment_version([e for e, _ in fl])      # written through raw pointers (see _FusedOptimizer.step)
        ints = [(e, m) for e, m in rest if e.numel() == 1 and not e.is_floating_point()]
        if ints:
            es = torch.stack([e.reshape(()) for e, _ in ints])
            ms = torch.stack([m.reshape(()) for _, m in ints])
            # src/ema.py:52: `decay * e + (1 - decay) * m` on integer tensors promotes to float32, then copy_ truncates
            new = (self.decay * es + (1. - self.decay) * ms).to(ints[0][0].dtype)
            torch._foreach_copy_([e.reshape(()) for e, _ in ints], list(new.unbind()))
        for e, m in rest:
            if not (e.numel() == 1 and not e.is_floating_point()):
                e.copy_(self.decay * e + (1. - self.decay) * m)

    def set(self, model):
        self._blend_unfused(model, 0.0)
