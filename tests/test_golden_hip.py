"""Golden vectors produced by the REFERENCE's own classes (tests/golden/make_golden.py) fed to the HIP
kernels themselves — through the C ABI in the rows layout, on the host simulator ('emu', CPU suite) and
on the MI355X ('gpu', pytest -m gpu).  fp32 kernels; tolerance 1e-3 of the tensor's max (north_star).

  gem_c16        GeneralizedMeanPooling fwd, dx, dp incl. the clamp path   -> mds_gem_fwd / mds_gem_bwd
  se3d           3D SqueezeExcite fwd/bwd                                  -> mds_se_fc_fwd / se_bwd_reduce / se_fc_bwd
  ir3d_c16_t5/t3 InvertedResidual3d: 2 train steps (BN buffers) + eval     -> the planner's IR block on every kernel
  tail_chain     forward_3d -> forward_head with grads wrt the features    -> module.forward_tail
"""
import numpy as np
import pytest
import torch

from backends import be, DT  # noqa: F401
from det_init import fill_deterministic, FakeEncoder
from mds import cabi
from mds.engine import Plan, Grad
from mds.structure import InvertedResidual3dP
from oracle import multidim_stacker_ref as orc
import mds


def T(a):
    return torch.from_numpy(np.asarray(a))


def close(got, want, tol=1e-3, msg=""):
    got, want = got.detach().float().cpu(), T(want).float() if not torch.is_tensor(want) else want.float()
    err = (got - want).abs().max().item()
    ref = max(want.abs().max().item(), 1e-12)
    assert got.shape == want.shape and err <= tol * ref, f"{msg}: max err {err:.3e} vs ref max {ref:.3e}"


def test_gem_reference_vectors(be, golden):
    d = golden("gem_c16")
    x = T(d["x"])                                    # (B, C, H, W), negative entries included
    B, C, H, W = x.shape
    rows = be.t(x.permute(0, 2, 3, 1).reshape(B * H * W, C))
    p = be.t(torch.tensor([3.0]))
    for split in (False, True):
        pooled = torch.empty(B, C, device=be.device)
        acc = torch.zeros(B, C, device=be.device, dtype=torch.float64) if split else None
        be.call("gem_fwd", cabi.make("mds_gem_fwd_args", dtype=0, groups=B, rows_per_group=H * W, C=C, y=rows,
                                     pro=cabi.pro(0), p=p, eps=1e-6, pooled=pooled, accum=acc))
        u = torch.empty_like(rows)
        dp = torch.zeros(1, device=be.device)
        acc2 = torch.zeros(B, C, device=be.device, dtype=torch.float64) if split else None
        be.call("gem_bwd", cabi.make("mds_gem_bwd_args", dtype=0, groups=B, rows_per_group=H * W, C=C, y=rows,
                                     pro=cabi.pro(0), p=p, eps=1e-6, pooled=pooled, dpooled=be.t(T(d["g"])), u=u, dp=dp,
                                     accum=acc2))
        be.sync()
        close(pooled, d["y"], msg="gem y")
        close(u.view(B, H, W, C).permute(0, 3, 1, 2), d["dx"], msg="gem dx")
        close(dp, d["dp"], 2e-3, msg="gem dp")


def test_se3d_reference_vectors(be, golden):
    d = golden("se3d")
    se = fill_deterministic(orc.SqueezeExcite(16, reduce_ratio=4, act_layer=torch.nn.SiLU), 3)   # weights of the fixture
    x, g = T(d["x"]), T(d["g"])                      # (B, C, T, H, W)
    B, C = x.shape[:2]
    rpg = x[0, 0].numel()
    R = se.conv_reduce.weight.shape[0]
    xr = be.t(x.permute(0, 2, 3, 4, 1).reshape(B * rpg, C))
    gr = be.t(g.permute(0, 2, 3, 4, 1).reshape(B * rpg, C))
    w1 = be.t(se.conv_reduce.weight.detach().reshape(R, C)); b1 = be.t(se.conv_reduce.bias.detach())
    w2 = be.t(se.conv_expand.weight.detach().reshape(C, R)); b2 = be.t(se.conv_expand.bias.detach())
    pooled = xr.view(B, rpg, C).mean(1).double().contiguous()       # x is the block's activation: its mean IS se_pool's output (fp64 buffer)
    hidden = torch.empty(B, R, device=be.device); gate = torch.empty(B, C, device=be.device)
    be.call("se_fc_fwd", cabi.make("mds_se_fc_fwd_args", groups=B, C=C, R=R, pooled=pooled, w1=w1, b1=b1, w2=w2, b2=b2,
                                   hidden=hidden, gate=gate, w2t=None))
    be.sync()
    y = xr.view(B, rpg, C) * gate[:, None, :]
    close(y.reshape(B, *x.shape[2:], C).permute(0, 4, 1, 2, 3), d["y"], msg="se y")
    dgate = torch.zeros(B, C, device=be.device, dtype=torch.float64)
    be.call("se_bwd_reduce", cabi.make("mds_se_bwd_reduce_args", dtype=0, groups=B, rows_per_group=rpg, C=C, u=gr, y=xr,
                                       scale=None, shift=None, dgate=dgate, mean=None, rstd=None, bnsums=None))
    dpooled = torch.empty(B, C, device=be.device)
    gw = [torch.zeros(R, C, device=be.device), torch.zeros(R, device=be.device), torch.zeros(C, R, device=be.device),
          torch.zeros(C, device=be.device)]
    be.call("se_fc_bwd", cabi.make("mds_se_fc_bwd_args", groups=B, C=C, R=R, rows_per_group=rpg, dgate=dgate, gate=gate,
                                   hidden=hidden, pooled=pooled, w1=w1, w2=w2, dpooled=dpooled,
                                   scratch=torch.empty(B, R, device=be.device), dw1=gw[0], db1=gw[1], dw2=gw[2], db2=gw[3],
                                   bnsums=None, bn_nblk=0, bn_stats=None, w2t=None))
    be.sync()
    dx = gr.view(B, rpg, C) * gate[:, None, :] + dpooled[:, None, :]
    close(dx.reshape(B, *x.shape[2:], C).permute(0, 4, 1, 2, 3), d["dx"], msg="se dx")
    close(gw[0].view(R, C, 1, 1, 1), d["grad.conv_reduce.weight"], msg="dw1")
    close(gw[1], d["grad.conv_reduce.bias"], msg="db1")
    close(gw[2].view(C, R, 1, 1, 1), d["grad.conv_expand.weight"], msg="dw2")
    close(gw[3], d["grad.conv_expand.bias"], msg="db2")


class BlockPlan(Plan):
    """The planner's inverted-residual block (every kernel family of the 3D tail) on its own."""

    def _build(self):
        blk, B, T_, H, W = self.m, self.B, self.T, self.H, self.W
        M = B * T_ * H * W
        self.xin = self.act(M, blk.cin)
        recs = self._recs["3d"]
        self.xout, _, _ = self._ir_block("f3d", recs, blk, blk.bn1.bn3d, blk.bn2.bn3d, blk.bn3.bn3d, self.xin, B, T_, H, W,
                                         1, B, True, False)
        if self.need_grad:
            self.dout = self.act(M, blk.cout)
            self.dxin = recs[-1]("b3d", Grad(self.dout), None).buf


def rows3d(x):            # (B, C, T, H, W) -> [B*T*H*W][C]
    return x.permute(0, 2, 3, 4, 1).reshape(-1, x.shape[1]).contiguous()


def unrows3d(r, shape):   # inverse
    B, C, T_, H, W = shape
    return r.view(B, T_, H, W, C).permute(0, 4, 1, 2, 3)


@pytest.mark.parametrize("tag", ["ir3d_c16_t5", "ir3d_c16_t3"])
def test_ir3d_reference_vectors(be, golden, tag):
    d = golden(tag)
    blk = fill_deterministic(InvertedResidual3dP(16, 16, 3, 4, 0.0), 26).to(be.device)
    x1 = T(d["x1"])
    B, C, T_, H, W = x1.shape
    plan = BlockPlan(blk, be.lib, be.device, "block", B, T_, H, W, 0, True, True, True)
    with plan.device_guard():
        # train step 1: output, input gradient, every parameter gradient, BN buffers
        plan.xin.tensor.copy_(be.t(rows3d(x1)).flatten())
        plan.begin_forward(None); plan.run("f3d")
        plan.dout.tensor.copy_(be.t(rows3d(T(d["g1"]))).flatten())
        plan.begin_backward(); plan.run("b3d"); plan.join_backward()
        be.sync()
        close(unrows3d(plan.xout.tensor.float().cpu(), x1.shape), d["y1"], msg="y1")
        close(unrows3d(plan.dxin.tensor.float().cpu(), x1.shape), d["dx1"], msg="dx1")
        flat = plan.grad_arena.tensor
        for n, p in blk.named_parameters():
            gw = flat[plan.poff[id(p)]:plan.poff[id(p)] + p.numel()].view(p.shape)
            want = T(d["grad1." + n])
            # conv biases feeding a train-mode BN have ~0 gradient: floor at 1e-2 of the typical scale
            close(gw, want, 2e-3 if want.abs().max() > 1e-4 else 1e9, msg="grad " + n)
        for n, b in blk.named_buffers():
            close(b.float(), d["buf1." + n], msg="buf1 " + n)
        # train step 2 (forward only): running statistics after two updates
        plan.xin.tensor.copy_(be.t(rows3d(T(d["x2"]))).flatten())
        plan.begin_forward(None); plan.run("f3d")
        be.sync()
        close(unrows3d(plan.xout.tensor.float().cpu(), x1.shape), d["y2"], msg="y2")
        for n, b in blk.named_buffers():
            close(b.float(), d["buf2." + n], msg="buf2 " + n)
    # eval mode: running statistics
    ev = BlockPlan(blk, be.lib, be.device, "block", B, T_, H, W, 0, False, False, False)
    with ev.device_guard():
        ev.xin.tensor.copy_(be.t(rows3d(T(d["x2"]))).flatten())
        ev.begin_forward(None); ev.run("f3d")
        be.sync()
        close(unrows3d(ev.xout.tensor.float().cpu(), x1.shape), d["y_eval"], msg="y_eval")


def test_tail_chain_reference_vectors(be, golden):
    d = golden("tail_chain")
    orc.ENCODER_REGISTRY["fake_grouping"] = FakeEncoder
    kw = dict(orc.BASIC_CONFIG_KWARGS, drop_rate=0.0, drop_path_rate=0.0)
    src = fill_deterministic(orc.MultiDimStacker(**dict(kw, model_name="fake_grouping")), 10, scale=0.15)   # the fixture's weights
    prod = mds.MultiDimStacker(**kw)
    tail = {k: v for k, v in src.state_dict().items() if not k.startswith("conv2d_encoder.")}
    missing, unexpected = prod.load_state_dict(tail, strict=False)
    assert not unexpected and all(k.startswith("conv2d_encoder.") for k in missing)
    prod = prod.to(be.device).train()
    if be.name == "emu":
        prod._lib = be.lib
    feats = be.t(T(d["feats"])).requires_grad_(True)
    logits = prod.forward_tail(feats)
    (logits * be.t(T(d["g"]))).sum().backward()
    be.sync()
    close(logits, d["logits"], msg="logits")
    close(feats.grad, d["dfeats"], 2e-3, msg="dfeats")
    named = dict(prod.named_parameters())
    for k in d:
        if k.startswith("grad."):
            close(named[k[5:]].grad, d[k], 2e-3, msg=k)
    assert all(p.grad is None for p in prod.conv2d_encoder.parameters())
