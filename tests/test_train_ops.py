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


@pytest.mark.parametrize("momentum,nesterov,dampening,wd", [(0.9, True, 0.0, 0.0), (0.9, False, 0.1, 1e-4), (0.0, False, 0.0, 5e-4)])
def test_fused_sgd_matches_torch_sgd(lib, momentum, nesterov, dampening, wd):
    """configs/ball_action/ball_finetune_long_004.py:51-55 builds ("SGD", {momentum 0.9, nesterov True})"""
    g = torch.Generator().manual_seed(5)
    shapes = [(576, 1, 3, 3, 3), (24,), (9000,), (3,), (2, 2816), (1,)]
    ref = [torch.randn(*s, generator=g).requires_grad_(True) for s in shapes]
    mine = [lib.t(p.detach().clone()).requires_grad_(True) for p in ref]
    kw = dict(lr=2e-2, momentum=momentum, nesterov=nesterov, dampening=dampening, weight_decay=wd)
    ro, mo = torch.optim.SGD(ref, **kw), train.FusedSGD(mine, **kw)
    total = sum(p.numel() for p in ref)
    for step in range(4):
        flat = torch.randn(total, generator=g)
        flat_d = lib.t(flat)
        off = 0
        for p, q in zip(ref, mine):
            n = p.numel()
            p.grad = flat[off:off + n].view_as(p).clone()
            q.grad = flat_d[off:off + n].view_as(q)
            off += n
        if step == 2:
            for grp in ro.param_groups + mo.param_groups:
                grp["lr"] = 5e-3
        ro.step(); mo.step()
    lib.sync()
    for p, q in zip(ref, mine):
        torch.testing.assert_close(q.detach().cpu(), p.detach(), rtol=2e-5, atol=2e-6)
    if momentum:
        sd = mo.state_dict()
        torch.testing.assert_close(sd["state"][2]["momentum_buffer"].cpu(), ro.state[ref[2]]["momentum_buffer"], rtol=2e-5, atol=1e-6)
        mo2 = train.FusedSGD(mine, **kw)
        mo2.load_state_dict(copy.deepcopy(sd))
        for p, q in zip(ref, mine):
            p.grad = torch.ones_like(p); q.grad = torch.ones_like(q)
        ro.step(); mo2.step()
        lib.sync()
        for p, q in zip(ref, mine):
            torch.testing.assert_close(q.detach().cpu(), p.detach(), rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("which", ["adamw", "sgd"])
def test_fused_optimizers_take_grad_scale_and_found_inf_on_the_device(lib, which):
    """the hand-shake of GradScaler.step with `_step_supports_amp_scaling` optimizers (src/argus_models.py:58-62):
    gradients arrive SCALED, `grad_scale` / `found_inf` are device tensors set as attributes for the call - the kernel
    unscales on load and skips the whole update when an overflow was found, with no host synchronisation"""
    g = torch.Generator().manual_seed(11)
    shapes = [(40, 7), (5000,), (3,)]
    ref = [torch.randn(*s, generator=g).requires_grad_(True) for s in shapes]
    mine = [lib.t(p.detach().clone()).requires_grad_(True) for p in ref]
    if which == "adamw":
        ro, mo = torch.optim.AdamW(ref, lr=1e-2), train.FusedAdamW(mine, lr=1e-2)
    else:
        ro, mo = torch.optim.SGD(ref, lr=1e-2, momentum=0.9, nesterov=True), train.FusedSGD(mine, lr=1e-2, momentum=0.9, nesterov=True)
    assert mo._step_supports_amp_scaling
    scale = 1024.0
    for step in range(3):
        overflow = step == 1
        for p, q in zip(ref, mine):
            gr = torch.randn(p.shape, generator=g)
            p.grad = gr.clone()
            q.grad = lib.t(gr * scale)
        before = [q.detach().clone() for q in mine]
        mo.grad_scale = lib.t(torch.tensor(scale))
        mo.found_inf = lib.t(torch.tensor(1.0 if overflow else 0.0))
        mo.step()
        del mo.grad_scale, mo.found_inf
        lib.sync()
        if overflow:
            for b, q in zip(before, mine):
                assert torch.equal(b, q.detach()), "a step with found_inf set must leave parameters untouched"
            continue                      # (the device step counter did not advance: next step's bias corrections are torch's)
        ro.step()
    for p, q in zip(ref, mine):
        torch.testing.assert_close(q.detach().cpu(), p.detach(), rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("which", ["adamw", "sgd"])
def test_a_step_skipped_on_found_inf_does_not_count(lib, which):
    """GradScaler skips the first fp16 steps (found_inf): torch's optimizers do not advance their step then, so the bias
    corrections (AdamW) and the first-step momentum initialisation (SGD with dampening) of the NEXT step are those of step 1.
    The counter lives on the device; state_dict() reports it."""
    g = torch.Generator().manual_seed(11)
    ref = [torch.randn(300, generator=g).requires_grad_(True), torch.randn(5000, generator=g).requires_grad_(True)]
    mine = [lib.t(p.detach().clone()).requires_grad_(True) for p in ref]
    if which == "adamw":
        ro = torch.optim.AdamW(ref, lr=1e-2, betas=(0.9, 0.99))
        mo = train.FusedAdamW(mine, lr=1e-2, betas=(0.9, 0.99))
    else:
        ro = torch.optim.SGD(ref, lr=1e-2, momentum=0.9, dampening=0.3)
        mo = train.FusedSGD(mine, lr=1e-2, momentum=0.9, dampening=0.3)
    versions = [q._version for q in mine]
    for step, inf in enumerate([1.0, 0.0, 1.0, 0.0, 0.0]):
        for p, q in zip(ref, mine):
            gr = torch.randn(p.shape, generator=g)
            p.grad = gr.clone(); q.grad = lib.t(gr)
        mo.found_inf = lib.t(torch.tensor([inf])); mo.grad_scale = lib.t(torch.tensor([1.0]))
        mo.step()
        if not inf:
            ro.step()
    lib.sync()
    for p, q in zip(ref, mine):
        torch.testing.assert_close(q.detach().cpu(), p.detach(), rtol=2e-5, atol=1e-6)
    assert float(mo.state_dict()["state"][0]["step"]) == 3.0
    assert all(q._version > v for q, v in zip(mine, versions)), "parameters written by the kernel must bump their version"


def test_model_ema_update_bumps_versions(lib):
    m = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.BatchNorm1d(4))
    m = m.to(lib.device)
    ema = train.ModelEma(m, decay=0.5)
    with torch.no_grad():
        m[0].weight.add_(1.0)
    v = ema.ema[0].weight._version
    ema.update(m)
    lib.sync()
    assert ema.ema[0].weight._version > v


def test_fused_optimizer_follows_parameter_reallocation(lib):
    """the device table is keyed by the parameters' addresses too (module.to() / p.data swap after a step)"""
    p = lib.t(torch.ones(300)).requires_grad_(True)
    opt = train.FusedSGD([p], lr=0.5)
    p.grad = torch.ones_like(p)
    opt.step()
    p.data = p.data.clone()            # new allocation, same Parameter object
    p.grad = torch.ones_like(p)
    opt.step()
    lib.sync()
    torch.testing.assert_close(p.detach().cpu(), torch.zeros(300))


def test_model_ema_without_float_entries_is_a_no_op(lib):
    class Counter(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.register_buffer("n", torch.tensor(5))
    m = Counter().to(lib.device)
    ema = train.ModelEma(m, decay=0.5)
    m.n += 4
    ema.update(m)
    assert int(ema.ema.n) == int(0.5 * 5 + 0.5 * 9)


def test_model_ema_held_on_another_device_follows_the_reference_arithmetic(lib):
    """ModelEma(model, device='cpu') (src/ema.py:37-55 with `device` set): update() and set() for float and integer entries equal
    `decay * e + (1 - decay) * m` entry by entry, bit for bit (the integer counter through float32 and a truncating copy)"""
    g = torch.Generator().manual_seed(9)
    m = torch.nn.Sequential(torch.nn.Linear(5, 4), torch.nn.BatchNorm1d(4)).to(lib.device)
    ema = train.ModelEma(m, decay=0.9, device="cpu")
    want = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    for step in range(3):
        with torch.no_grad():
            for prm in m.parameters():
                prm.add_(torch.randn(prm.shape, generator=g).to(prm.device))
            m[1].running_mean.add_(0.25)
            m[1].num_batches_tracked.add_(7)
        ema.update(m)
        for k, v in m.state_dict().items():
            blend = 0.9 * want[k] + (1. - 0.9) * v.detach().cpu()
            want[k] = blend.to(want[k].dtype)
    lib.sync()
    got = ema.ema.state_dict()
    assert all(v.device.type == "cpu" for v in got.values())
    for k in want:
        assert torch.equal(got[k], want[k]), k
    ema.set(m)
    for k, v in m.state_dict().items():
        assert torch.equal(ema.ema.state_dict()[k], v.detach().cpu()), k


@pytest.mark.gpu
def test_train_ops_run_on_a_non_default_stream():
    """a trainer that owns its HIP stream: loss, both optimizers and the EMA are launched on torch's CURRENT stream (64-bit handle)"""
    if not torch.cuda.is_available():
        pytest.skip("needs the GPU")
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    x = torch.randn(64, 2, generator=g); t = (torch.rand(64, 2, generator=g) < 0.5).float()
    xr = x.clone().requires_grad_(True)
    lr = orc.sigmoid_focal_loss(xr, t, alpha=-1.0, gamma=1.2, reduction="mean"); lr.backward()
    own = torch.cuda.Stream(dev)
    for which in ("adamw", "sgd"):
        ps = [torch.nn.Parameter(torch.randn(33, 7, generator=g).to(dev)), torch.nn.Parameter(torch.randn(129, generator=g).to(dev))]
        rs = [torch.nn.Parameter(p.detach().cpu().clone()) for p in ps]
        opt = train.FusedAdamW(ps, lr=1e-2) if which == "adamw" else train.FusedSGD(ps, lr=1e-2, momentum=0.9, nesterov=True)
        ropt = torch.optim.AdamW(rs, lr=1e-2) if which == "adamw" else torch.optim.SGD(rs, lr=1e-2, momentum=0.9, nesterov=True)
        net = torch.nn.Linear(7, 3).to(dev)
        ema_ref = copy.deepcopy(net).cpu()
        with torch.cuda.stream(own):
            xd = x.to(dev).requires_grad_(True)
            lp = train.FocalLoss(-1.0, 1.2, "mean")(xd, t.to(dev)); lp.backward()
            for _ in range(3):
                for p, r in zip(ps, rs):
                    gr = torch.randn(p.shape, generator=g)
                    p.grad = gr.to(dev); r.grad = gr.clone()
                opt.step(); ropt.step()
            ema = train.ModelEma(net, decay=0.9)
            with torch.no_grad():
                for p in net.parameters():
                    p.add_(1.0)
            ema.update(net)
        own.synchronize()
        torch.testing.assert_close(lp.detach().cpu(), lr.detach(), rtol=2e-5, atol=1e-6)
        torch.testing.assert_close(xd.grad.cpu(), xr.grad, rtol=2e-5, atol=1e-7)
        for p, r in zip(ps, rs):
            torch.testing.assert_close(p.detach().cpu(), r.detach(), rtol=1e-5, atol=1e-6)
        for (k, e), (_, m0) in zip(ema.ema.state_dict().items(), ema_ref.state_dict().items()):
            torch.testing.assert_close(e.cpu(), 0.9 * m0 + 0.1 * (m0 + 1.0), rtol=1e-5, atol=1e-6)
