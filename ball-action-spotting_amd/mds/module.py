"""Drop-in ``MultiDimStacker`` for MI355X.

Same constructor, attributes, methods and ``state_dict`` as the reference class
(``/root/reference/src/models/multidim_stacker.py:137-243``); ``src/argus_models.py``,
``src/predictors.py`` and ``src/ema.py`` run on it unchanged.  All arithmetic
runs in the hand-written gfx950 kernels of ``libmds_hip.so`` (C ABI: include/mds.h) driven by
``engine.Plan``; if the library is missing this module raises — there is no eager/CPU fallback.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from . import cabi
from .engine import Plan
from .structure import (EncoderP, InvertedResidual3dP, GeneralizedMeanPoolingP, TAIL_BN_EPS)


class _PlanCache:
    """Per-module cache of launch plans.  Never copied or pickled (EMA deep-copies the module,
    src/ema.py:40): a copy starts with an empty cache and re-plans on first use."""

    def __init__(self):
        self.plans = {}

    def __deepcopy__(self, memo):
        return _PlanCache()

    def __reduce__(self):
        return (_PlanCache, ())


class _Release:
    """Marks a plan reusable when the autograd graph that references it dies."""

    def __init__(self, plan):
        self.plan = plan

    def __del__(self):
        self.plan.in_flight = False


class _MDSFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, module, plan, *params):
        ctx.plan, ctx.module = plan, module
        ctx.token = _Release(plan)
        plan.in_flight = True
        plan.bind_input(x)
        plan.begin_forward(module._mask_override)
        plan.run("f2d"); plan.run("f3d"); plan.run("fhead")
        B = x.shape[0]
        return plan.logits.tensor.view(B, -1).clone()

    @staticmethod
    def backward(ctx, dlogits):
        plan = ctx.plan
        plan.dlogits.tensor.copy_(dlogits.reshape(-1).float())
        plan.begin_backward()
        plan.run("bhead"); plan.run("b3d"); plan.run("b2d")
        plan.join_backward()
        flat = plan.grad_arena.tensor.clone()      # one launch; the arena is reused next step
        sync = getattr(ctx.module, "_grad_sync", None)
        if sync is not None:
            sync(flat)                             # data parallel: one RCCL all-reduce of the flat buffer
        grads = []
        for p in plan.params:
            if p.requires_grad:
                off = plan.poff[id(p)]
                grads.append(flat[off:off + p.numel()].view(p.shape))
            else:
                grads.append(None)
        plan.in_flight = False
        return (None, None, None, *grads)


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
        # pretrained=True would pull timm/tf_efficientnetv2_b0.in1k from the HF hub; offline the caller
        # loads weights (scripts/ball_action/train.py:48-62 does exactly that for every later stage).
        self.conv2d_encoder = EncoderP(in_chans=stack_size, drop_path_rate=drop_path_rate)
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

    # ------------------------------------------------------------------ plumbing
    def _apply(self, fn, *a, **k):
        self._cache = _PlanCache()            # .to()/.cuda()/.half() move parameters: re-plan
        return super()._apply(fn, *a, **k)

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
            return cabi.MDS_BF16              # fp16 autocast (the reference's AMP) also maps to bf16 storage
        return cabi.MDS_F32

    def _plan(self, x, kind, B, T, H, W, need_grad):
        lib = self._library(x)
        enc_grad = any(p.requires_grad for p in self.conv2d_encoder.parameters())
        key = (kind, B, T, H, W, self._code(), self.training, need_grad, enc_grad, x.device)
        pool = self._cache.plans.setdefault(key, [])
        for plan in pool:
            if not plan.in_flight and not plan.stale():
                return plan
        pool[:] = [p for p in pool if not p.stale()]
        plan = Plan(self, lib, x.device, kind, B, T, H, W, self._code(), self.training, need_grad, enc_grad)
        pool.append(plan)
        return plan

    # ------------------------------------------------------------------ reference API
    def forward(self, x):
        b, t, h, w = x.shape
        assert t == self.num_frames and t % self.stack_size == 0
        x = x.float().contiguous()
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        plan = self._plan(x, "full", b, t, h, w, need_grad)
        if need_grad:
            return _MDSFunction.apply(x, self, plan, *plan.params)
        plan.bind_input(x)
        plan.begin_forward(self._mask_override)
        plan.run("f2d"); plan.run("f3d"); plan.run("fhead")
        return plan.logits.tensor.view(b, -1).clone()

    def _inference_only(self, what):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError(f"mds: {what} called on its own is an inference path (src/predictors.py:50-72); "
                                      f"wrap it in torch.no_grad() — training goes through forward()")

    def forward_2d(self, x):
        self._inference_only("forward_2d")
        b, t, h, w = x.shape
        assert t % self.stack_size == 0
        s = t // self.stack_size
        x = x.float().contiguous()
        plan = self._plan(x, "2d", b, t, h, w, False)
        plan.bind_input(x)
        plan.begin_forward(None)
        plan.run("f2d")
        f = plan.feat.tensor.view(b, s, plan.h, plan.w, self.num_3d_features)
        return f.permute(0, 1, 4, 2, 3).float().contiguous()      # (b, S, 192, h, w) like the reference

    def forward_3d(self, x):
        self._inference_only("forward_3d")
        b, t, c, h, w = x.shape
        assert c == self.num_3d_features and t == self.num_stacks
        plan = self._plan(x, "3d", b, t * self.stack_size, h, w, False)
        plan.feat.tensor.view(b, t, h, w, c).copy_(x.permute(0, 1, 3, 4, 2))
        plan.begin_forward(None)
        plan.run("f3d")
        cq = self.num_features // t
        y = plan.out3d.tensor.view(b, t, h, w, cq)
        return y.permute(0, 1, 4, 2, 3).reshape(b, self.num_features, h, w).float().contiguous()

    def forward_head(self, x):
        self._inference_only("forward_head")
        b, f, h, w = x.shape
        t = self.num_stacks
        cq = f // t
        plan = self._plan(x, "head", b, t * self.stack_size, h, w, False)
        plan.yq.tensor.view(b, t, h, w, cq).copy_(x.view(b, t, cq, h, w).permute(0, 1, 3, 4, 2))
        plan.begin_forward(self._mask_override)
        plan.run("fhead")
        return plan.logits.tensor.view(b, -1).clone()
