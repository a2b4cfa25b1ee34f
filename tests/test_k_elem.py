"""Row-streaming and latency-class kernels against torch fp32 references (autograd for bwd)."""
import pytest
import torch
import torch.nn.functional as F

from backends import be, DT, assert_close  # noqa: F401
from mds import cabi

SLOTS = cabi.MDS_STAT_SLOTS


def gen(s):
    return torch.Generator().manual_seed(s)


def bn_finalize(be, stats, count, gamma, beta, eps=1e-3, momentum=0.1, training=1, rm=None, rv=None, nbt=None):
    C = gamma.numel()
    out = torch.empty(4, C, device=be.device)
    be.call("bn_finalize", cabi.make("mds_bn_finalize_args", C=C, count=count, stats=stats, gamma=gamma,
                                     beta=beta, eps=eps, momentum=momentum, training=training,
                                     running_mean=rm, running_var=rv, num_batches_tracked=nbt, out=out))
    return out


def test_bn_finalize_train_and_eval(be):
    C, M = 24, 1000
    y = torch.randn(M, C, generator=gen(0)) * 2 + 0.5
    stats = torch.zeros(SLOTS, 2, C, dtype=torch.float64)
    stats[3, 0] = y[:400].sum(0); stats[3, 1] = (y[:400] ** 2).sum(0)
    stats[17, 0] = y[400:].sum(0); stats[17, 1] = (y[400:] ** 2).sum(0)
    gamma = 1 + 0.1 * torch.randn(C, generator=gen(1)); beta = 0.1 * torch.randn(C, generator=gen(2))
    rm = be.t(torch.zeros(C)); rv = be.t(torch.ones(C)); nbt = be.t(torch.zeros((), dtype=torch.long))
    out = bn_finalize(be, be.t(stats), M, be.t(gamma), be.t(beta), eps=1e-3, rm=rm, rv=rv, nbt=nbt)
    be.sync()
    bn = torch.nn.BatchNorm1d(C, eps=1e-3)
    with torch.no_grad():
        bn.weight.copy_(gamma); bn.bias.copy_(beta)
    ref = bn(y)
    z = y * out[0].cpu() + out[1].cpu()
    assert_close(z, ref, "f32", msg="train normalise")
    assert_close(rm, bn.running_mean, "f32"); assert_close(rv, bn.running_var, "f32")
    assert int(nbt.item()) == 1
    out = bn_finalize(be, None, 0, be.t(gamma), be.t(beta), eps=1e-3, training=0, rm=rm, rv=rv)
    be.sync()
    bn.eval()
    assert_close(y * out[0].cpu() + out[1].cpu(), bn(y), "f32", msg="eval normalise")


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("M,C,act,use_mask,use_sc", [(300, 16, 0, True, True), (77, 112, 1, False, False),
                                                       (64, 1152, 0, True, True), (1000, 48, 0, False, True)])
def test_bn_res(be, dt, M, C, act, use_mask, use_sc):
    code, tdt = DT[dt]
    g = gen(M + C)
    rpg = 40
    y = torch.randn(M, C, generator=g).to(tdt); sc_ = torch.randn(M, C, generator=g).to(tdt)
    scale = 1 + 0.2 * torch.randn(C, generator=g); shift = 0.2 * torch.randn(C, generator=g)
    mask = (torch.rand((M + rpg - 1) // rpg, generator=g) > 0.3).float() / 0.7
    out = torch.empty(M, C, dtype=tdt, device=be.device)
    be.call("bn_res", cabi.make("mds_bn_res_args", dtype=code, M=M, C=C, y=be.t(y), scale=be.t(scale),
                                shift=be.t(shift), act=act, mask=be.t(mask) if use_mask else None,
                                rows_per_group=rpg, shortcut=be.t(sc_) if use_sc else None, out=out))
    be.sync()
    z = y.float() * scale + shift
    if act:
        z = F.silu(z)
    if use_mask:
        z = z * mask[torch.arange(M) // rpg, None]
    if use_sc:
        z = z + sc_.float()
    assert_close(out, z, dt)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("packed,C,RD", [(True, 48, 12), (False, 48, 12), (True, 328, 28), (True, 1152, 48),
                                         (True, 1536, 64), (True, 2048, 20)])   # > 1280 channels: two trips of the mat-vec; R = 64: 16 rows per wave
def test_se_forward_backward(be, dt, packed, C, RD):
    """se_pool + se_fc_fwd + (gated consumer) and the SE backward chain vs autograd; with and without
    the packed [R][C] copy of w2 (made by pack_weights / MDS_PACK_IO_F32)."""
    code, tdt = DT[dt]
    G, R_ = 3, 70
    M = G * R_
    g = gen(5)
    y = torch.randn(M, C, generator=g).to(tdt)
    scale = 1 + 0.2 * torch.randn(C, generator=g); shift = 0.2 * torch.randn(C, generator=g)
    w1 = torch.randn(RD, C, generator=g) * 0.3; b1 = torch.randn(RD, generator=g) * 0.1
    w2 = torch.randn(C, RD, generator=g) * 0.3; b2 = torch.randn(C, generator=g) * 0.1
    u = torch.randn(M, C, generator=g).to(tdt)          # grad wrt gated output
    # reference
    yf = y.float().requires_grad_(True)
    p = {k: v.clone().requires_grad_(True) for k, v in dict(w1=w1, b1=b1, w2=w2, b2=b2).items()}
    a = F.silu(yf * scale + shift).view(G, R_, C)
    pooled_ref = a.mean(1)
    hid = pooled_ref @ p["w1"].t() + p["b1"]
    gate_ref = torch.sigmoid(F.silu(hid) @ p["w2"].t() + p["b2"])
    out = a * gate_ref[:, None, :]
    a.retain_grad()
    (out * u.float().view(G, R_, C)).sum().backward()
    # kernels
    yd = be.t(y); ud = be.t(u); scd = be.t(scale); shd = be.t(shift)
    pooled = torch.zeros(G, C, device=be.device, dtype=torch.float64)      # cross-block sums that feed activations: fp64
    be.call("se_pool", cabi.make("mds_se_pool_args", dtype=code, groups=G, rows_per_group=R_, C=C, y=yd,
                                 scale=scd, shift=shd, pooled=pooled))
    hidden = torch.empty(G, RD, device=be.device); gate = torch.empty(G, C, device=be.device)
    w1d, b1d, w2d, b2d = be.t(w1), be.t(b1), be.t(w2), be.t(b2)
    w2t = None
    if packed:
        w2t = torch.full((RD, C), float("nan"), device=be.device)
        job = cabi.make("mds_pack_job", src=w2d, dst=w2t, kind=cabi.MDS_PACK_IO_F32, O=C, I=RD, taps=1)
        raw = torch.frombuffer(bytearray(bytes(job)), dtype=torch.uint8).to(be.device)
        be.lib.check(be.lib.fn["pack_weights"](raw.data_ptr(), 1, C * RD, code, be.stream()), "pack_weights")
        be.sync()
        assert torch.equal(w2t.cpu(), w2.t().contiguous())
    be.call("se_fc_fwd", cabi.make("mds_se_fc_fwd_args", groups=G, C=C, R=RD, pooled=pooled, w1=w1d, b1=b1d,
                                   w2=w2d, b2=b2d, hidden=hidden, gate=gate, w2t=w2t))
    dgate = torch.zeros(G, C, device=be.device, dtype=torch.float64)
    be.call("se_bwd_reduce", cabi.make("mds_se_bwd_reduce_args", dtype=code, groups=G, rows_per_group=R_, C=C,
                                       u=ud, y=yd, scale=scd, shift=shd, dgate=dgate))
    dpooled = torch.empty(G, C, device=be.device)
    dw1 = torch.zeros(RD, C, device=be.device); db1 = torch.zeros(RD, device=be.device)
    dw2 = torch.zeros(C, RD, device=be.device); db2 = torch.zeros(C, device=be.device)
    be.call("se_fc_bwd", cabi.make("mds_se_fc_bwd_args", groups=G, C=C, R=RD, rows_per_group=R_, dgate=dgate,
                                   gate=gate, hidden=hidden, pooled=pooled, w1=w1d, w2=w2d, dpooled=dpooled,
                                   scratch=torch.empty(G, RD, device=be.device), dw1=dw1, db1=db1, dw2=dw2, db2=db2,
                                   w2t=w2t))
    be.sync()
    assert_close(pooled, pooled_ref, dt, msg="pooled")
    assert_close(gate, gate_ref, dt, msg="gate")
    assert_close(dw1, p["w1"].grad, dt, scale=10, msg="dw1"); assert_close(db1, p["b1"].grad, dt, scale=10, msg="db1")
    assert_close(dw2, p["w2"].grad, dt, scale=10, msg="dw2"); assert_close(db2, p["b2"].grad, dt, scale=10, msg="db2")
    # total grad wrt a = u*gate + dpooled(broadcast): check through the BN-backward g-source
    da_ref = a.grad.view(M, C)
    da = ud.float().cpu() * gate.cpu()[torch.arange(M) // R_] + dpooled.cpu()[torch.arange(M) // R_]
    assert_close(da, da_ref, dt, scale=3, msg="da")


def test_se_parameter_gradients_table_equals_the_per_layer_launches(be):
    """mds_se_fc_bwd_params_table (a device-resident array of layers, grid.y = layer: what the planner issues once per gradient
    bucket) against one mds_se_fc_bwd_params launch per layer, on layers of different widths: bit-identical, accumulating (+=)"""
    g = gen(77)
    G = 5
    SJ = cabi.STRUCTS["mds_se_fc_bwd_args"]
    shapes = [(48, 12), (1152, 48), (328, 28)]
    jobs, outs = [], []
    for C, R in shapes:
        dgate = be.t(torch.randn(G, C, generator=g).double()); gate = be.t(torch.rand(G, C, generator=g))
        hidden = be.t(torch.randn(G, R, generator=g)); pooled = be.t(torch.randn(G, C, generator=g).double())
        scratch = be.t(torch.randn(G, R, generator=g)); w1 = be.t(torch.randn(R, C, generator=g)); w2 = be.t(torch.randn(C, R, generator=g))
        init = [torch.randn(R, C, generator=g), torch.randn(R, generator=g), torch.randn(C, R, generator=g), torch.randn(C, generator=g)]
        two = []
        for _ in range(2):
            gr = [be.t(t.clone()) for t in init]
            two.append((gr, cabi.make("mds_se_fc_bwd_args", groups=G, C=C, R=R, rows_per_group=70, dgate=dgate, gate=gate, hidden=hidden,
                                      pooled=pooled, w1=w1, w2=w2, dpooled=torch.empty(G, C, device=be.device), scratch=scratch,
                                      dw1=gr[0], db1=gr[1], dw2=gr[2], db2=gr[3])))
        be.call("se_fc_bwd_params", two[0][1])
        jobs.append(two[1][1]); outs.append((two[0][0], two[1][0]))
    arr = (SJ * len(jobs))(*jobs)
    tab = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(be.device)
    be.call("se_fc_bwd_params_table", cabi.make("mds_se_fc_bwd_table_args", jobs=tab, njobs=len(jobs), max_rc=max(C * R for C, R in shapes)))
    be.sync()
    for (C, R), (ref, got) in zip(shapes, outs):
        for a_, b_, name in zip(ref, got, ("dw1", "db1", "dw2", "db2")):
            assert torch.equal(a_.cpu(), b_.cpu()), (C, R, name)
    with pytest.raises(Exception):
        be.call("se_fc_bwd_params_table", cabi.make("mds_se_fc_bwd_table_args", jobs=tab, njobs=0, max_rc=1))


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("mode,C", [(0, 40), (1, 40), (2, 40), (3, 40), (1, 1152)])
def test_bn_backward_chain(be, dt, mode, C):
    """reduce -> finalize -> apply == autograd through act(batch_norm(y)) for every g-source
    (C = 1152 takes the two-slice path of the reduce kernels)."""
    code, tdt = DT[dt]
    M, rpg = 330, 110
    G = M // rpg
    g = gen(11 + mode)
    y = (torch.randn(M, C, generator=g) * 1.5 + 0.3).to(tdt)
    u = torch.randn(M, C, generator=g).to(tdt)
    gamma = 1 + 0.2 * torch.randn(C, generator=g); beta = 0.2 * torch.randn(C, generator=g)
    gate = torch.rand(G, C, generator=g); dpool = 0.1 * torch.randn(G, C, generator=g)
    mask = torch.tensor([0.0, 1.25, 1.25])
    eps = 1e-5
    yf = y.float().requires_grad_(True)
    gam = gamma.clone().requires_grad_(True); bet = beta.clone().requires_grad_(True)
    z = F.batch_norm(yf, None, None, gam, bet, True, 0.1, eps)
    grp = torch.arange(M) // rpg
    uf = u.float()
    if mode == 0:
        z.backward(uf)
    elif mode == 1:
        F.silu(z).backward(uf)
    elif mode == 2:
        F.silu(z).backward(uf * gate[grp] + dpool[grp])
    else:
        (z * mask[grp, None]).backward(uf)
    # kernels
    stats = torch.zeros(SLOTS, 2, C, dtype=torch.float64)
    stats[0, 0] = y.float().sum(0); stats[0, 1] = (y.float() ** 2).sum(0)
    bn = bn_finalize(be, be.t(stats), M, be.t(gamma), be.t(beta), eps=eps)
    gs = cabi.gsrc(mode, be.t(u), be.t(gate), be.t(dpool), be.t(mask), rpg)
    st2 = torch.zeros(SLOTS, 2, C, device=be.device, dtype=torch.float64)     # backward sums: fp64 slots
    yd = be.t(y)
    be.call("bn_bwd_reduce", cabi.make("mds_bn_bwd_reduce_args", dtype=code, M=M, C=C, g=gs, y=yd, bn=bn, stats=st2))
    dgamma = torch.zeros(C, device=be.device); dbeta = torch.zeros(C, device=be.device)
    coef = torch.empty(3, C, device=be.device)
    be.call("bn_bwd_finalize", cabi.make("mds_bn_bwd_finalize_args", C=C, count=M, stats=st2, gamma=be.t(gamma),
                                         bn=bn, dgamma=dgamma, dbeta=dbeta, coef=coef, batch_stats=1))
    dy = torch.empty(M, C, dtype=tdt, device=be.device)
    be.call("bn_bwd_apply", cabi.make("mds_bn_bwd_apply_args", dtype=code, M=M, C=C, g=gs, y=yd, bn=bn, coef=coef, dy=dy))
    be.sync()
    assert_close(dgamma, gam.grad, dt, scale=20, msg="dgamma")
    assert_close(dbeta, bet.grad, dt, scale=20, msg="dbeta")
    assert_close(dy, yf.grad, dt, scale=2, msg="dy")


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("pro_mode,split,R_,C", [(0, False, 35, 16), (2, False, 35, 16), (2, True, 35, 16), (2, True, 920, 64)])
def test_gem_fwd_bwd(be, dt, pro_mode, split, R_, C, golden):
    """single-block path and the row-split path (caller-zeroed accumulators)"""
    code, tdt = DT[dt]
    G = 4
    g = gen(21)
    y = (torch.randn(G * R_, C, generator=g) * 1.5).to(tdt)
    scale = 1 + 0.2 * torch.randn(C, generator=g); shift = 0.2 * torch.randn(C, generator=g)
    p = torch.tensor([3.0]); dpo = torch.randn(G, C, generator=g)
    yf = y.float().requires_grad_(True); pp = p.clone().requires_grad_(True)
    a = F.silu(yf * scale + shift) if pro_mode else yf
    a.retain_grad()
    pooled_ref = a.view(G, R_, C).clamp(min=1e-6).pow(pp).mean(1).pow(1.0 / pp)
    (pooled_ref * dpo).sum().backward()
    yd = be.t(y)
    pro = cabi.pro(pro_mode, be.t(scale), be.t(shift))
    pooled = torch.empty(G, C, device=be.device); pd = be.t(p)
    acc1 = torch.zeros(G, C, device=be.device, dtype=torch.float64) if split else None
    acc2 = torch.zeros(G, C, device=be.device, dtype=torch.float64) if split else None
    be.call("gem_fwd", cabi.make("mds_gem_fwd_args", dtype=code, groups=G, rows_per_group=R_, C=C, y=yd, pro=pro,
                                 p=pd, eps=1e-6, pooled=pooled, accum=acc1))
    u = torch.empty(G * R_, C, dtype=tdt, device=be.device); dp = torch.zeros(1, device=be.device)
    be.call("gem_bwd", cabi.make("mds_gem_bwd_args", dtype=code, groups=G, rows_per_group=R_, C=C, y=yd, pro=pro,
                                 p=pd, eps=1e-6, pooled=pooled, dpooled=be.t(dpo), u=u, dp=dp, accum=acc2))
    be.sync()
    assert_close(pooled, pooled_ref, dt, msg="pooled")
    assert_close(u, a.grad, dt, msg="u")
    assert_close(dp, pp.grad, dt, scale=5, msg="dp")


def test_gem_matches_reference_golden(be, golden):
    """GeM against the vectors produced by the reference class itself (tests/golden/gem.npz)."""
    d = golden("gem")
    x = torch.from_numpy(d["x"])                       # (2,6,5,7) NCHW
    B, C, H, W = x.shape
    Cp = 8                                             # pad channels to the kernel's multiple of 8
    rows = torch.zeros(B * H * W, Cp)
    rows[:, :C] = x.permute(0, 2, 3, 1).reshape(-1, C)
    rows[:, C:] = 1.0
    pooled = torch.empty(B, Cp, device=be.device); p = be.t(torch.tensor([3.0]))
    yd = be.t(rows)
    be.call("gem_fwd", cabi.make("mds_gem_fwd_args", dtype=0, groups=B, rows_per_group=H * W, C=Cp, y=yd,
                                 pro=cabi.pro(0), p=p, eps=1e-6, pooled=pooled))
    dpo = torch.zeros(B, Cp); dpo[:, :C] = torch.from_numpy(d["g"])
    u = torch.empty(B * H * W, Cp, device=be.device); dp = torch.zeros(1, device=be.device)
    be.call("gem_bwd", cabi.make("mds_gem_bwd_args", dtype=0, groups=B, rows_per_group=H * W, C=Cp, y=yd,
                                 pro=cabi.pro(0), p=p, eps=1e-6, pooled=pooled, dpooled=be.t(dpo), u=u, dp=dp))
    be.sync()
    assert_close(pooled[:, :C], torch.from_numpy(d["y"]), "f32", msg="y")
    dx = u.cpu()[:, :C].view(B, H, W, C).permute(0, 3, 1, 2)
    assert_close(dx, torch.from_numpy(d["dx"]), "f32", msg="dx")
    assert_close(dp, torch.from_numpy(d["dp"]), "f32", scale=5, msg="dp")


def test_head_fwd_bwd(be):
    B, Fdim, NC = 3, 200, 5
    g = gen(31)
    pooled = torch.randn(B, Fdim, generator=g); w = torch.randn(NC, Fdim, generator=g) * 0.1
    b = torch.randn(NC, generator=g); mask = (torch.rand(B, Fdim, generator=g) > 0.2).float() / 0.8
    dl = torch.randn(B, NC, generator=g)
    pf = pooled.clone().requires_grad_(True); wf = w.clone().requires_grad_(True); bf = b.clone().requires_grad_(True)
    ref = F.linear(pf * mask, wf, bf)
    ref.backward(dl)
    logits = torch.empty(B, NC, device=be.device)
    pd, md, wd = be.t(pooled), be.t(mask), be.t(w)
    be.call("head_fwd", cabi.make("mds_head_fwd_args", B=B, F=Fdim, NC=NC, pooled=pd, mask=md, w=wd, b=be.t(b), logits=logits))
    dpo = torch.empty(B, Fdim, device=be.device); dw = torch.zeros(NC, Fdim, device=be.device); db = torch.zeros(NC, device=be.device)
    be.call("head_bwd", cabi.make("mds_head_bwd_args", B=B, F=Fdim, NC=NC, pooled=pd, mask=md, w=wd, dlogits=be.t(dl),
                                  dpooled=dpo, dw=dw, db=db))
    be.sync()
    assert_close(logits, ref, "f32"); assert_close(dpo, pf.grad, "f32")
    assert_close(dw, wf.grad, "f32"); assert_close(db, bf.grad, "f32")


@pytest.mark.parametrize("tta", [1, 2, 3])
def test_head_fwd_probs(be, tta):
    """the predictor's nn.Sigmoid + mean over the TTA group (src/predictors.py:69-70) from the head's own launch"""
    B, Fdim, NC = 6, 333, 2
    g = gen(32 + tta)
    pooled = torch.randn(B, Fdim, generator=g); w = torch.randn(NC, Fdim, generator=g) * 0.1; b = torch.randn(NC, generator=g)
    ref = F.linear(pooled, w, b)
    logits = torch.empty(B, NC, device=be.device); probs = torch.empty(B // tta, NC, device=be.device)
    be.call("head_fwd", cabi.make("mds_head_fwd_args", B=B, F=Fdim, NC=NC, pooled=be.t(pooled), mask=None, w=be.t(w), b=be.t(b),
                                  logits=logits, probs=probs, tta=tta))
    be.sync()
    assert_close(logits, ref, "f32")
    assert_close(probs, torch.sigmoid(ref).view(B // tta, tta, NC).mean(1), "f32")


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_pack_weights(be, dt):
    import ctypes as C_
    code, tdt = DT[dt]
    g = gen(41)
    w3 = torch.randn(20, 12, 3, 3, generator=g); w1 = torch.randn(24, 16, 1, 1, generator=g)
    ws = torch.randn(32, 3, 3, 3, generator=g)
    srcs = [be.t(w3), be.t(w3), be.t(w1), be.t(ws)]
    dsts = [torch.empty(20 * 9 * 12, dtype=tdt, device=be.device), torch.empty(12 * 9 * 20, dtype=tdt, device=be.device),
            torch.empty(24 * 16, dtype=tdt, device=be.device), torch.empty(32 * 32, dtype=tdt, device=be.device)]
    kinds = [(0, 20, 12, 9), (1, 20, 12, 9), (0, 24, 16, 1), (2, 32, 27, 1)]
    Job = cabi.STRUCTS["mds_pack_job"]
    jobs = (Job * 4)()
    for j, (s, d, (k, O, I, t)) in enumerate(zip(srcs, dsts, kinds)):
        jobs[j].src = s.data_ptr(); jobs[j].dst = d.data_ptr(); jobs[j].kind = k; jobs[j].O = O; jobs[j].I = I; jobs[j].taps = t
    raw = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(be.device)
    be.lib.check(be.lib.fn["pack_weights"](raw.data_ptr(), 4, 20 * 12 * 9, code, be.stream()), "pack_weights")
    be.sync()
    assert_close(dsts[0].view(20, 9, 12), w3.view(20, 12, 9).permute(0, 2, 1), dt)
    assert_close(dsts[1].view(12, 9, 20), w3.view(20, 12, 9).flip(2).permute(1, 2, 0), dt)
    assert_close(dsts[2].view(24, 16), w1.view(24, 16), dt)
    ref = torch.zeros(32, 32); ref[:, :27] = ws.view(32, 27)
    assert_close(dsts[3].view(32, 32), ref, dt)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("O,I", [(24, 16), (70, 44), (192, 1152), (33, 5)])
def test_pack_weights_transposed(be, dt, O, I):
    """[O][I] -> [I][O] packs (MDS_PACK_IO_FLIP with one tap: the 1x1 data-gradient filters; MDS_PACK_IO_F32: the fp32
    squeeze-excite w2 copy) go through 32 x 32 LDS tiles: ragged edges, several tiles per block."""
    code, tdt = DT[dt]
    w = torch.randn(O, I, generator=gen(43))
    src = be.t(w)
    d_flip = torch.empty(I * O, dtype=tdt, device=be.device)
    d_f32 = torch.empty(I * O, dtype=torch.float32, device=be.device)
    Job = cabi.STRUCTS["mds_pack_job"]
    jobs = (Job * 2)()
    for j, (d, k) in enumerate(((d_flip, cabi.MDS_PACK_IO_FLIP), (d_f32, cabi.MDS_PACK_IO_F32))):
        jobs[j].src = src.data_ptr(); jobs[j].dst = d.data_ptr(); jobs[j].kind = k; jobs[j].O = O; jobs[j].I = I; jobs[j].taps = 1
    raw = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(be.device)
    be.lib.check(be.lib.fn["pack_weights"](raw.data_ptr(), 2, O * I, code, be.stream()), "pack_weights")
    be.sync()
    assert_close(d_flip.view(I, O), w.t(), dt)
    assert torch.equal(d_f32.view(I, O).cpu(), w.t().contiguous())
