"""SURVEY 8(f) N2 — fused focal loss, multi-tensor AdamW and EMA (mds.train) against the oracle's restatement of
src/losses.py, torch.optim.AdamW and the reference's ModelEma arithmetic (src/ema.py:47-55)."""
import copy

import pytest
import torch

from backends import be  # noqa: F401
from oracle import multidim_stacker_ref as orc
from mds import train


@pytest.fixture
def lib(be):
    train.LIB = be.lib if be.name == "emu" else None
    yield be
    train.LIB = None


@pytest.mark.parametrize("alpha,gamma,reduction,shape", [(-1.0, 1.2, "mean", (4, 2)), (0.25, 2.0, "sum", (3, 15)), (-1.0, 2.0, "none", (5, 7)),
                                                         (0.6, 1.2, "mean", (1000, 3))])
def test_focal_loss_value_and_gradient(lib, alpha, gamma, reduction, shape):
    g = torch.Generator().manual_seed(7)
    x = (3 * torch.randn(*shape, generator=g))
    x.view(-1)[0] = 30.0; x.view(-1)[1] = -30.0          # saturated logits: the stable BCE form must hold
    t = (torch.rand(*shape, generator=g) < 0.4).float()
    t.view(-1)[0] = 0.0; t.view(-1)[1] = 1.0
    w = torch.randn(*shape, generator=g) if reduction == "none" else torch.tensor(1.7)
    xr = x.clone().requires_grad_(True)
    lr = orc.sigmoid_focal_loss(xr, t, alpha=alpha, gamma=gamma, reduction=reduction)
    (lr * w).sum().backward()
    xd = lib.t(x).requires_grad_(True)
    lp = train.FocalLoss(alpha, gamma, reduction)(xd, lib.t(t))
    (lp * lib.t(w)).sum().backward()
    lib.sync()
    assert lp.shape == lr.shape
    torch.testing.assert_close(lp.detach().cpu(), lr.detach(), rtol=2e-5, atol=1e-6)
    torch.testing.assert_close(xd.grad.cpu(), xr.grad, rtol=2e-5, atol=1e-7)


def test_fused_adamw_matches_torch_adamw(lib):
    g = torch.Generator().manual_seed(3)
    shapes = [(32, 3, 3, 3), (32,), (5000,), (7,), (192, 48, 1, 1), (1,), (2, 1280)]     # > one chunk, odd sizes, scalars
    ref = [torch.randn(*s, generator=g).requires_grad_(True) for s in shapes]
    mine = [lib.t(p.detach().clone()).requires_grad_(True) for p in ref]
    ro = torch.optim.AdamW(ref, lr=3e-3, betas=(0.9, 0.99), eps=1e-7, weight_decay=0.05)
    mo = train.FusedAdamW(mine, lr=3e-3, betas=(0.9, 0.99), eps=1e-7, weight_decay=0.05)
    total = sum(p.numel() for p in ref)
    for step in range(4):
        flat = torch.randn(total, generator=g)
        flat_d = lib.t(flat)                                          # gradients handed over as views of ONE buffer
        off = 0
        for p, q in zip(ref, mine):
            n = p.numel()
            p.grad = flat[off:off + n].view_as(p).clone()
            q.grad = flat_d[off:off + n].view_as(q) if step != 2 else flat_d[off:off + n].view_as(q).clone()   # step 2: separate tensors
            off += n
        if step == 3:
            for grp in ro.param_groups + mo.param_groups:
                grp["lr"] = 1e-3                                      # schedulers change lr in place
        ro.step(); mo.step()
    lib.sync()
    for p, q in zip(ref, mine):
        torch.testing.assert_close(q.detach().cpu(), p.detach(), rtol=2e-5, atol=2e-6)
    sd = mo.state_dict()
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"} and float(sd["state"][0]["step"]) == 4.0
    torch.testing.assert_close(sd["state"][2]["exp_avg"].cpu(), ro.state[ref[2]]["exp_avg"], rtol=2e-5, atol=1e-7)
    mo2 = train.FusedAdamW(mine, lr=1e-3, betas=(0.9, 0.99), eps=1e-7, weight_decay=0.05)
    mo2.load_state_dict(copy.deepcopy(sd))
    for p, q in zip(ref, mine):
        p.grad = torch.ones_like(p); q.grad = torch.ones_like(q)
    ro.step(); mo2.step()
    lib.sync()
    for p, q in zip(ref, mine):
        torch.testing.assert_close(q.detach().cpu(), p.detach(), rtol=2e-5, atol=2e-6)


def test_model_ema_matches_reference_arithmetic(lib):
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.BatchNorm2d(8), torch.nn.Linear(8, 5000))
    ref_ema = copy.deepcopy(net).eval()
    net_d = copy.deepcopy(net).to(lib.device)
    ema = train.ModelEma(net_d, decay=0.9)
    for it in range(3):
        with torch.no_grad():
            for (_, a), (_, b) in zip(net.state_dict().items(), net_d.state_dict().items()):
                delta = torch.randn(a.shape) if a.is_floating_point() else torch.tensor(3)
                a.add_(delta.to(a.dtype)); b.add_(delta.to(b.dtype).to(b.device))
            for e, m in zip(ref_ema.state_dict().values(), net.state_dict().values()):      # src/ema.py:47-55
                e.copy_(0.9 * e + (1. - 0.9) * m)
        ema.update(net_d)
    lib.sync()
    for (k, e), r in zip(ema.ema.state_dict().items(), ref_ema.state_dict().values()):
        torch.testing.assert_close(e.cpu(), r, rtol=1e-5, atol=1e-6, msg=k)
    assert not ema.ema.training
