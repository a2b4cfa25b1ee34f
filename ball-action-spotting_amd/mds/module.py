"""Drop-in ``MultiDimStacker`` for MI355X.

Same constructor, attributes, methods and ``state_dict`` as the reference class
(``/root/reference/src/models/multidim_stacker.py:137-243``); ``src/argus_models.py``,
``src/predictors.py`` and ``src/ema.py`` run on it unchanged.  All arithmetic
runs in the hand-written gfx950 kernels of ``libmds_hip.so`` (C ABI: include/mds.h) driven by
``engine.Plan``; if the library is missing this module raises — there is no eager/CPU fallback.

``torch.compile(module)`` (``scripts/ball_action/train.py:83-86``): the training forward is a registered operator,
``torch.ops.mds.forward`` (``torch.library.custom_op`` with a fake implementation and an autograd formula whose backward is
the operator ``torch.ops.mds.backward``), so Dynamo / AOTAutograd / Inductor trace THROUGH ``forward`` with
``fullgraph=True`` and see one opaque node each way - the hot path already is a static launch schedule, there is nothing for a
tracing compiler to add.  The operator finds its module through an integer handle (operators take tensors and scalars,
not modules); BatchNorm's running statistics are updated by the launch as a side effect, as in eager mode.
``MDS_CUSTOM_OP=0`` selects the former ``torch.autograd.Function`` behind ``torch.compiler.disable`` (a graph break).
"""
from __future__ import annotations

import copy
import itertools
import os
import warnings
import weakref
from collections import OrderedDict
from typing import List, Optional, Tuple

import torch
from torch import nn

from . import cabi
from .engine import Plan
from .structure import (EncoderP, InvertedResidual3dP, GeneralizedMeanPoolingP, TAIL_BN_EPS)

MAX_PLANS = 8     # launch plans kept per module (LRU); each pins its activation arena (12 GB at config 2)


class _PlanCache:
    """Per-module LRU cache of launch plans.  Never copied or pickled (EMA deep-copies the module,
    src/ema.py:40): a copy starts with an empty cache and re-plans on first use."""

    def __init__(self):
        self.plans = OrderedDict()       # key -> [Plan, ...]

    def __deepcopy__(self, memo):
        return _PlanCache()

    def __reduce__(self):
        return (_PlanCache, ())

    def count(self):
        return sum(len(v) for v in self.plans.values())

    def idle(self):
        return sum(1 for v in self.plans.values() for p in v if not p.in_flight)

    def evict(self, keep_key):
        """drop least-recently-used IDLE plans until at most MAX_PLANS of them remain.  Plans that are in flight - held by
        an autograd graph, or owned by a StreamPredictor for its lifetime - do not count against the budget: they cannot be
        dropped, and counting them made every later shape of the module re-plan on each call once a predictor held eight."""
        for key in list(self.plans):
            if self.idle() <= MAX_PLANS:
                break
            if key == keep_key:
                continue
            pool = self.plans[key]
            pool[:] = [p for p in pool if p.in_flight]
            if not pool:
                del self.plans[key]


class _Release:
    """Marks a plan reusable when the autograd graph that references it dies."""

    def __init__(self, plan):
        self.plan = plan

    def __del__(self):
        plan = getattr(self, "plan", None)      # __del__ may run on a half-built / torn-down object
        if plan is not None:
            plan.in_flight = False


class _Release2:
    """the same for the registered operator's path: lives on the autograd context, releases the plan named by the token"""

    def __init__(self, handle, token):
        self.handle, self.token = handle, token

    def __del__(self):
        try:
            _release_token(self.handle, self.token)
        except Exception:       # interpreter shutdown
            pass


class _MDSFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, module, plan, *params):
        ctx.plan, ctx.module = plan, module
        ctx.token = _Release(plan)
        ctx.consumed = False
        ctx.save_for_backward(x)           # version-checked: an in-place edit of x before backward raises
        plan.in_flight = True
        plan.generation += 1
        ctx.generation = plan.generation
        with plan.device_guard():
            plan.bind_input(x)
            plan.begin_forward(module._mask_override)
            plan.run("f2d"); plan.run("f3d"); plan.run("fhead")
            B = x.shape[0]
            return plan.logits.tensor.view(B, -1).clone()

    @staticmethod
    def backward(ctx, dlogits):
        plan = ctx.plan
        if ctx.consumed or ctx.generation != plan.generation:
            raise RuntimeError("mds: the activations of this forward have been released (backward already ran, or the "
                               "plan was reused by a later forward); a second backward / retain_graph is not supported")
        (x,) = ctx.saved_tensors            # raises if x was modified in place since forward
        ctx.consumed = True
        flat = _run_backward(ctx.module, plan, x, dlogits)
        grads = []
        for p in plan.params:
            if p.requires_grad:
                off = plan.poff[id(p)]
                grads.append(flat[off:off + p.numel()].view(p.shape))
            else:
                grads.append(None)
        plan.in_flight = False
        return (None, None, None, *grads)


class _TailFunction(torch.autograd.Function):
    """forward_3d -> forward_head as one differentiable call on given 2D features (fine-tuning the temporal
    tail on cached features; also how the reference-generated `tail_chain` vectors reach the kernels)."""

    @staticmethod
    def forward(ctx, feats, module, plan, *params):
        ctx.plan, ctx.token = plan, _Release(plan)
        plan.in_flight = True
        plan.generation += 1
        ctx.generation, ctx.shape = plan.generation, feats.shape
        b, s, c, h, w = feats.shape
        with plan.device_guard():
            plan.feat.tensor.view(b, s, h, w, c).copy_(feats.detach().permute(0, 1, 3, 4, 2))
            plan.begin_forward(module._mask_override)
            plan.run("f3d"); plan.run("fhead")
            return plan.logits.tensor.view(b, -1).clone()

    @staticmethod
    def backward(ctx, dlogits):
        plan = ctx.plan
        if ctx.generation != plan.generation or not plan.in_flight:
            raise RuntimeError("mds: the activations of this forward have been released")
        b, s, c, h, w = ctx.shape
        with plan.device_guard():
            plan.bind_dlogits(dlogits)
            plan.begin_backward()
            plan.run("bhead"); plan.run("b3d")
            plan.join_backward()
            flat = plan.grad_arena.tensor.clone()
            dfeats = plan.dfeat.tensor.view(b, s, h, w, c).permute(0, 1, 4, 2, 3).float().contiguous()
        grads = [flat[plan.poff[id(p)]:plan.poff[id(p)] + p.numel()].view(p.shape) if p.requires_grad else None
                 for p in plan.tail_params]
        plan.in_flight = False
        return (dfeats, None, None, *grads)


class _SubFunction(torch.autograd.Function):
    """forward_2d / forward_3d / forward_head called on their own with autograd on (the reference's are ordinary
    differentiable methods, multidim_stacker.py:210-237): one plan of that kind with its backward schedule; gradients for
    the input (3d, head) and for the parameters of that part of the network only."""

    @staticmethod
    def forward(ctx, inp, module, plan, kind, *params):
        ctx.plan, ctx.kind, ctx.token, ctx.module = plan, kind, _Release(plan), module
        ctx.params = params
        plan.in_flight = True
        plan.generation += 1
        ctx.generation, ctx.shape = plan.generation, inp.shape
        ctx.save_for_backward(inp)
        return module._run_sub(plan, kind, inp)

    @staticmethod
    def backward(ctx, dout):
        plan, kind = ctx.plan, ctx.kind
        if ctx.generation != plan.generation or not plan.in_flight:
            raise RuntimeError("mds: the activations of this forward have been released")
        (inp,) = ctx.saved_tensors
        m = ctx.module
        dinp = None
        with plan.device_guard():
            plan.begin_backward()
            if kind == "2d":
                b, t, h, w = ctx.shape
                s_ = t // m.stack_size
                plan.bind_input(inp)
                plan.dfeat_in.tensor.view(b, s_, plan.h, plan.w, m.num_3d_features).copy_(dout.permute(0, 1, 3, 4, 2))
                plan.run("b2d")
            elif kind == "3d":
                b, t, c, h, w = ctx.shape
                cq = m.num_features // t
                plan.uq_in.tensor.view(b, t, h, w, cq).copy_(dout.view(b, t, cq, h, w).permute(0, 1, 3, 4, 2))
                plan.run("b3d")
                dinp = plan.dfeat.tensor.view(b, t, h, w, c).permute(0, 1, 4, 2, 3).float().contiguous()
            else:
                b, f, h, w = ctx.shape
                t = m.num_stacks
                plan.bind_dlogits(dout)
                plan.run("bhead")
                dinp = plan.dyq_out.tensor.view(b, t, h, w, f // t).permute(0, 1, 4, 2, 3).reshape(b, f, h, w).float().contiguous()
            plan.join_backward()
            flat = plan.grad_arena.tensor.clone()
        grads = [flat[plan.poff[id(p)]:plan.poff[id(p)] + p.numel()].view(p.shape) if p.requires_grad else None for p in ctx.params]
        plan.in_flight = False
        return (dinp, None, None, None, *grads)


# ------------------------------------------------------------------------------------------------ registered operators
_MODULES = weakref.WeakValueDictionary()      # handle -> MultiDimStacker (operators cannot take a module argument)
_HANDLES = itertools.count(1)
_TOKENS = itertools.count(1)
USE_CUSTOM_OP = os.environ.get("MDS_CUSTOM_OP", "1") != "0"


def _module_of(handle: int):
    m = _MODULES.get(handle)
    if m is None:
        raise RuntimeError(f"mds: module handle {handle} is gone (the MultiDimStacker it named was deleted)")
    return m


# A training forward also updates the BatchNorm running statistics and draws DropPath / dropout masks.  torch refuses an autograd
# formula on an operator that declares mutated arguments, so the operator stays functional in its schema and carries the
# `nondeterministic_seeded` tag instead (it IS seeded-random with drop rates > 0): graph passes that merge or reorder pure nodes
# leave tagged operators alone.  tests/test_module_emu.py runs two identical calls in one compiled graph and checks that the
# statistics advance twice.
@torch.library.custom_op("mds::forward", mutates_args=(), tags=(torch.Tag.nondeterministic_seeded,))
def _op_forward(x: torch.Tensor, params: List[torch.Tensor], handle: int, need_grad: bool, code: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """logits of MultiDimStacker.forward + a CPU token naming the launch plan that holds the activations for backward.
    `params` only tells autograd what the result depends on - the kernels read the module's parameters in place."""
    m = _module_of(handle)
    b, t, h, w = x.shape
    plan = m._plan(x, "full", b, t, h, w, need_grad, code=code)
    token = 0
    if need_grad:
        live = m._live
        while len(live) >= 4:                      # forwards whose backward never came (loss evaluated under grad mode, then dropped)
            live.pop(next(iter(live))).in_flight = False
        token = next(_TOKENS)
        live[token] = plan
        plan.in_flight = True
        plan.generation += 1
    with plan.device_guard():
        plan.bind_input(x)
        plan.begin_forward(m._mask_override)
        plan.run("f2d"); plan.run("f3d"); plan.run("fhead")
        return plan.logits.tensor.view(b, -1).clone(), torch.tensor(token, dtype=torch.int64)


@_op_forward.register_fake
def _(x, params, handle, need_grad, code):
    return x.new_empty((x.shape[0], _module_of(handle).classifier.out_features), dtype=torch.float32), torch.empty((), dtype=torch.int64)


@torch.library.custom_op("mds::backward", mutates_args=())
def _op_backward(dlogits: torch.Tensor, x: torch.Tensor, token: torch.Tensor, handle: int) -> torch.Tensor:
    """the flat fp32 gradient arena (all parameters, parameter order) of the forward named by `token`"""
    m = _module_of(handle)
    plan = m._live.pop(int(token.item()), None)          # (a CPU scalar: no device synchronisation)
    if plan is None:
        raise RuntimeError("mds: the activations of this forward have been released (backward already ran, or the "
                           "plan was reused by a later forward); a second backward / retain_graph is not supported")
    flat = _run_backward(m, plan, x, dlogits)
    plan.in_flight = False
    return flat


@_op_backward.register_fake
def _(dlogits, x, token, handle):
    return dlogits.new_empty((sum(p.numel() for p in _module_of(handle).parameters()),), dtype=torch.float32)


def _release_token(handle, token):
    """the autograd node of a grad-enabled forward died without running backward (loss only evaluated, an exception): its plan
    - a full activation arena - becomes reusable at once instead of after four later forwards"""
    m = _MODULES.get(handle)
    if m is not None:
        plan = m._live.pop(token, None)
        if plan is not None:
            plan.in_flight = False


def _op_setup_context(ctx, inputs, output):
    x, params, handle, need_grad, code = inputs
    ctx.save_for_backward(x, output[1])     # x is version-checked: an in-place edit before backward raises
    ctx.handle = handle
    tok = output[1]
    if need_grad and type(tok) is torch.Tensor and not torch.compiler.is_compiling():     # (not while tracing: fake tensors have no value)
        ctx._mds_release = _Release2(handle, int(tok))
    ctx.meta = [(p.numel(), p.shape, p.requires_grad) for p in params]


def _op_backward_formula(ctx, dlogits, dtoken):
    x, token = ctx.saved_tensors
    flat = torch.ops.mds.backward(dlogits.contiguous(), x, token, ctx.handle)
    grads, off = [], 0
    for n, shape, req in ctx.meta:
        grads.append(flat[off:off + n].view(shape) if req else None)
        off += n
    return None, grads, None, None, None


_op_forward.register_autograd(_op_backward_formula, setup_context=_op_setup_context)


def _run_backward(module, plan, x, dlogits):
    """issue the backward schedule of `plan`; returns the flat gradient buffer (averaged over ranks under data parallelism)"""
    sync = getattr(module, "_grad_sync", None)
    bucketed = hasattr(sync, "on_cut")
    with plan.device_guard():
        plan.bind_input(x)
        plan.bind_dlogits(dlogits)
        plan.begin_backward()
        plan.cut_hook = sync.on_cut if bucketed else None     # data parallel: all-reduce each slice as soon as it is final
        plan.run("bhead"); plan.run("b3d"); plan.run("b2d")
        plan.cut_hook = None
        plan.join_backward()
        if bucketed:
            world = sync.finish(plan)
            flat = plan.grad_arena.tensor / world if world > 1 else plan.grad_arena.tensor.clone()
        else:
            flat = plan.grad_arena.tensor.clone()  # one launch; the arena is reused next step
            if sync is not None:
                sync(flat)                         # data parallel: one RCCL all-reduce of the flat buffer
    return flat


def _load_pretrained_encoder(encoder: nn.Module, model_name: str, in_chans: int):
    """``pretrained=True`` (configs/ball_action/sampling_weights_001.py:36): the reference gets ImageNet
    weights from ``timm.create_model(..., pretrained=True)`` (multidim_stacker.py:166-176).  Do the same when
    timm and its weight cache are reachable; otherwise say so LOUDLY — never a silent random init."""
    try:
        import timm  # noqa: F401  (absent from the build image; present in the reference's environment)
        src = timm.create_model(model_name, pretrained=True, in_chans=in_chans, features_only=True, out_indices=[4])
        sd = src.state_dict()
        own = encoder.state_dict()
        if set(sd) != set(own) or any(sd[k].shape != own[k].shape for k in own):
            raise RuntimeError("timm state_dict does not match the mds encoder layout (timm version != 0.9.2?)")
        encoder.load_state_dict(sd)
        return True
    except Exception as e:  # ImportError, no network / no cached weights, layout mismatch
        msg = (f"mds.MultiDimStacker(pretrained=True): ImageNet weights for '{model_name}' could not be loaded "
               f"({type(e).__name__}: {e}).")
        if os.environ.get("MDS_ALLOW_RANDOM_INIT") != "1":
            # the reference's timm.create_model(pretrained=True) fails hard here too; stage-1 training from a random encoder
            # is never what the caller meant
            raise RuntimeError(msg + " Install timm with its cached weights, construct with pretrained=False and load a "
                               "checkpoint (load_state_dict / src.utils.load_weights_from_pretrain), or set "
                               "MDS_ALLOW_RANDOM_INIT=1 to continue with a RANDOMLY INITIALISED encoder.") from e
        warnings.warn(msg + " MDS_ALLOW_RANDOM_INIT=1: the 2D encoder is RANDOMLY INITIALISED.", RuntimeWarning, stacklevel=3)
        return False


class MultiDimStacker(nn.Module):
    def __init__(self, model_name: str, num_classes: int, num_frames: int = 15, stack_size: int = 3,
                 index_2d_features: int = 4, pretrained: bool = False, num_3d_blocks: int = 2,
                 num_3d_features: int = 192, num_3d_stack_proj: int = 256, expansion_3d_ratio: int = 6,
                 se_reduce_3d_ratio: int = 24, drop_rate: float = 0., drop_path_rate: float = 0.,
                 act_layer: str = "silu", **kwargs):
        super().__init__()
        assert num_frames > 0 and num_frames % stack_size == 0
        if model_name.split(".")[0] != "tf_efficientnetv2_b0":
            raise NotImplementedError(f"mds HIP engine implements the reference's encoder tf_efficientnetv2_b0, got {model_name}")
        if act_layer != "silu" or index_2d_features != 4 or stack_size != 3:
            raise NotImplementedError("mds HIP engine: act_layer='silu', index_2d_features=4, stack_size=3 (all reference configs)")
        self.num_frames = num_frames
        self.stack_size = stack_size
        self.num_3d_features = num_3d_features
        self.num_stacks = num_frames // stack_size
        self.num_features = num_3d_stack_proj * self.num_stacks
        self.drop_rate = drop_rate
        self.conv2d_encoder = EncoderP(in_chans=stack_size, drop_path_rate=drop_path_rate)
        self.pretrained_loaded = _load_pretrained_encoder(self.conv2d_encoder, model_name, stack_size) if pretrained else False
        enc_chs = self.conv2d_encoder.feature_info[index_2d_features]["num_chs"]
        self.conv2d_projection = nn.Sequential(
            nn.Conv2d(enc_chs, num_3d_features, 1, bias=False), nn.BatchNorm2d(num_3d_features, eps=TAIL_BN_EPS))
        self.conv3d_encoder = nn.Sequential(*[
            InvertedResidual3dP(num_3d_features, num_3d_features, expansion_3d_ratio, se_reduce_3d_ratio, drop_path_rate)
            for _ in range(num_3d_blocks)])
        self.conv3d_projection = nn.Sequential(
            nn.Conv2d(num_3d_features, num_3d_stack_proj, 1, bias=False), nn.BatchNorm2d(num_3d_stack_proj, eps=TAIL_BN_EPS))
        self.global_pool = GeneralizedMeanPoolingP(3.0)
        self.classifier = nn.Linear(self.num_features, num_classes, bias=True)
        # engine state (not part of state_dict)
        self.compute_dtype = "auto"           # "auto": bf16 under autocast, fp32 otherwise | "bf16" | "f32"
        self._cache = _PlanCache()
        self._lib: Optional[cabi.Lib] = None  # tests inject the kernel simulator here; product: cabi.load()
        self._mask_override = None            # parity tests: host-supplied DropPath/dropout masks
        self._grad_sync = None                # mds.parallel.data_parallel installs the all-reduce here
        self._register()

    def _register(self):
        """a fresh operator handle for this instance (never shared with the module it was copied / unpickled from)"""
        self._handle = next(_HANDLES)
        self._live = {}                       # token -> plan of a grad-enabled forward whose backward has not run yet
        self._warned_fp16 = False
        _MODULES[self._handle] = self

    def __deepcopy__(self, memo):             # src/ema.py:40 deep-copies the module: the copy is its own operator target
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = {} if k == "_live" else (v if k == "_lib" else copy.deepcopy(v, memo))     # (a loaded library is shared)
        new._register()
        return new

    def __getstate__(self):                   # plans of forwards in flight are not part of the module's state
        state = dict(self.__dict__)
        state["_live"] = {}
        return state

    def __setstate__(self, state):            # torch.save(module) / pickle
        super().__setstate__(state)
        self._register()

    # ------------------------------------------------------------------ plumbing
    def _apply(self, fn, *a, **k):
        self._cache = _PlanCache()            # .to()/.cuda()/.half() move parameters: re-plan
        return super()._apply(fn, *a, **k)

    def clear_plans(self):
        """Release every cached launch plan (and the activation arenas they pin)."""
        self._cache = _PlanCache()

    def _library(self, x):
        if self._lib is not None:
            return self._lib
        if not x.is_cuda:
            raise cabi.MdsError("MultiDimStacker (mds) runs on MI355X only: move the module and input to cuda")
        return cabi.load()

    def _code(self):
        if self.compute_dtype == "bf16":
            return cabi.MDS_BF16
        if self.compute_dtype == "f32":
            return cabi.MDS_F32
        if torch.is_autocast_enabled() or torch.is_autocast_enabled('cpu'):
            # The reference's AMP is fp16 autocast + GradScaler (src/argus_models.py:36,54).  The kernels store bf16: same
            # range as fp32 (a GradScaler then never finds an overflow and simply keeps its scale), 3 mantissa bits fewer
            # than fp16 - said once, not silently.
            dt = torch.get_autocast_dtype("cuda" if torch.is_autocast_enabled() else "cpu")
            if dt == torch.float16 and not self._warned_fp16 and not torch.compiler.is_compiling():
                self._warned_fp16 = True
                warnings.warn("mds.MultiDimStacker: fp16 autocast is executed with bf16 storage (fp32 accumulation); "
                              "torch.autocast(dtype=torch.bfloat16) is the matching setting and needs no GradScaler.", stacklevel=3)
            return cabi.MDS_BF16
        return cabi.MDS_F32

    def _plan(self, x, kind, B, T, H, W, need_grad, ingest=None, code=None):
        lib = self._library(x)
        enc_grad = any(p.requires_grad for p in self.conv2d_encoder.parameters())
        code = self._code() if code is None else code
        key = (kind, B, T, H, W, code, self.training, need_grad, enc_grad, x.device, ingest)
        cache = self._cache
        pool = cache.plans.setdefault(key, [])
        cache.plans.move_to_end(key)
        for plan in pool:
            if not plan.in_flight and not plan.stale():
                return plan
        pool[:] = [p for p in pool if not p.stale()]
        plan = Plan(self, lib, x.device, kind, B, T, H, W, code, self.training, need_grad, enc_grad, ingest=ingest)
        pool.append(plan)
        cache.evict(key)
        return plan

    # ------------------------------------------------------------------ reference API
    def forward(self, x):
        b, t, h, w = x.shape
        assert t == self.num_frames and t % self.stack_size == 0
        x = x.float().contiguous()
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if USE_CUSTOM_OP:      # the registered operator: traceable by torch.compile(fullgraph=True), one opaque node each way
            return torch.ops.mds.forward(x, list(self.parameters()), self._handle, need_grad, self._code())[0]
        return self._forward_untraced(x, need_grad)

    @torch.compiler.disable
    def _forward_untraced(self, x, need_grad):
        b, t, h, w = x.shape
        plan = self._plan(x, "full", b, t, h, w, need_grad)
        if need_grad:
            return _MDSFunction.apply(x, self, plan, *plan.params)
        with plan.device_guard():
            plan.bind_input(x)
            plan.begin_forward(self._mask_override)
            plan.run("f2d"); plan.run("f3d"); plan.run("fhead")
            return plan.logits.tensor.view(b, -1).clone()

    SUB_PARAMS = {"2d": ("conv2d_encoder.", "conv2d_projection."), "3d": ("conv3d_encoder.", "conv3d_projection."),
                  "head": ("global_pool.", "classifier.")}

    def _sub(self, kind, inp, B, T, H, W):
        """one of the three sub-forwards: inference plan under no_grad (src/predictors.py:58-70), differentiable otherwise"""
        names = self.SUB_PARAMS[kind]
        params = [p for n, p in self.named_parameters() if n.startswith(names)]
        need_grad = torch.is_grad_enabled() and (inp.requires_grad or any(p.requires_grad for p in params))
        plan = self._plan(inp, kind, B, T, H, W, need_grad)
        if need_grad:
            return _SubFunction.apply(inp, self, plan, kind, *params)
        return self._run_sub(plan, kind, inp)

    def _run_sub(self, plan, kind, x):
        with plan.device_guard():
            if kind == "2d":
                b, t, h, w = x.shape
                plan.bind_input(x)
                plan.begin_forward(self._mask_override)
                plan.run("f2d")
                f = plan.feat.tensor.view(b, t // self.stack_size, plan.h, plan.w, self.num_3d_features)
                return f.permute(0, 1, 4, 2, 3).float().contiguous()      # (b, S, 192, h, w) like the reference
            if kind == "3d":
                b, t, c, h, w = x.shape
                plan.feat.tensor.view(b, t, h, w, c).copy_(x.detach().permute(0, 1, 3, 4, 2))
                plan.begin_forward(self._mask_override)
                plan.run("f3d")
                cq = self.num_features // t
                y = plan.out3d.tensor.view(b, t, h, w, cq)
                return y.permute(0, 1, 4, 2, 3).reshape(b, self.num_features, h, w).float().contiguous()
            b, f, h, w = x.shape
            t = self.num_stacks
            cq = f // t
            plan.yq.tensor.view(b, t, h, w, cq).copy_(x.detach().view(b, t, cq, h, w).permute(0, 1, 3, 4, 2))
            plan.begin_forward(self._mask_override)
            plan.run("fhead")
            return plan.logits.tensor.view(b, -1).clone()

    @torch.compiler.disable
    def forward_2d(self, x):
        b, t, h, w = x.shape
        assert t % self.stack_size == 0
        return self._sub("2d", x.float().contiguous(), b, t, h, w)

    @torch.compiler.disable
    def forward_3d(self, x):
        b, t, c, h, w = x.shape
        assert c == self.num_3d_features and t == self.num_stacks
        return self._sub("3d", x, b, t * self.stack_size, h, w)

    @torch.compiler.disable
    def forward_tail(self, feats):
        """logits = forward_head(forward_3d(feats)) for feats of shape (b, num_stacks, 192, h, w), differentiable
        with respect to the features and the tail parameters (not part of the reference API)."""
        b, t, c, h, w = feats.shape
        assert c == self.num_3d_features and t == self.num_stacks
        plan = self._plan(feats, "tail", b, t * self.stack_size, h, w, True)
        plan.tail_params = [p for n, p in self.named_parameters() if not n.startswith("conv2d_")]
        return _TailFunction.apply(feats, self, plan, *plan.tail_params)

    @torch.compiler.disable
    def forward_head(self, x):
        b, f, h, w = x.shape
        return self._sub("head", x, b, self.num_stacks * self.stack_size, h, w)
