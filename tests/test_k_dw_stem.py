"""Depthwise (2D/3D) and stem kernels against torch conv / autograd references."""
import pytest
import torch
import torch.nn.functional as F

from backends import be, DT, assert_close  # noqa: F401
from mds import cabi, geometry as geo

SLOTS = cabi.MDS_STAT_SLOTS


def gen(s):
    return torch.Generator().manual_seed(s)


def to_rows(x):   # (N,C,T,H,W) -> [N][T][H][W][C]
    return x.permute(0, 2, 3, 4, 1).contiguous()


def dw_ref(a, w, stride, kt, pads):
    """a: (N,C,T,H,W) activated input, w: (C,1,kt,3,3)."""
    (pt, pb), (pl, pr) = pads
    tp = 1 if kt == 3 else 0
    a = F.pad(a, (pl, pr, pt, pb, tp, tp))
    return F.conv3d(a, w, None, (1, stride, stride), 0, 1, a.shape[1])


DW_CASES = [  # N,T,H,W,C,stride,kt
    (2, 1, 9, 13, 48, 1, 1),
    (1, 1, 14, 37, 72, 1, 1),     # several bands/segments, half-filled channel chunk
    (3, 1, 5, 8, 136, 1, 1),
    (1, 1, 12, 18, 16, 2, 1),     # even: pad 0/1
    (2, 1, 11, 15, 24, 2, 1),     # odd: pad 1/1
    (1, 1, 12, 15, 16, 2, 1),     # mixed: pad_t 0, pad_l 1
    (1, 1, 11, 38, 72, 2, 1),     # mixed: pad_t 1, pad_l 0; several segments
    (2, 3, 5, 7, 24, 1, 3),
    (1, 5, 6, 9, 576, 1, 3),
    (2, 11, 6, 13, 72, 1, 3),     # the 33-frame configuration's stack length: time chunks of 4 / 4 / 3 slices (dw3g_*)
    (1, 4, 5, 9, 24, 1, 3),       # exactly one chunk
    (1, 9, 7, 8, 136, 1, 3),      # 4 / 4 / 1
    (3, 2, 5, 6, 16, 1, 3),
]


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("N,T,H,W,C,stride,kt", DW_CASES)
def test_dw_fwd_bwd(be, dt, N, T, H, W, C, stride, kt):
    _dw_fwd_bwd(be, dt, N, T, H, W, C, stride, kt)


@pytest.mark.parametrize("length", [5, 7, 13])
@pytest.mark.parametrize("N,T,H,W,C,stride,kt", [(1, 1, 14, 37, 72, 1, 1), (2, 5, 6, 23, 136, 1, 3)])
def test_dw_strip_lengths(be, length, N, T, H, W, C, stride, kt):
    """the launchers pick the strip length from the block count (dw2_len / dw3_len); any length gives the same result:
    forced lengths with ragged last segments, several segments per row"""
    knob = cabi.MDS_KNOB_DW2_L if kt == 1 else cabi.MDS_KNOB_DW3_L
    be.lib.check(be.lib.fn["dev_set"](knob, length), "dev_set")
    try:
        _dw_fwd_bwd(be, "bf16", N, T, H, W, C, stride, kt)
    finally:
        be.lib.fn["dev_set"](knob, 0)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("N,T,H,W,C,stride,kt", [c for c in DW_CASES if c[5] == 1 and c[6] == 1])
def test_dw_fwd_six_row_bands(be, dt, N, T, H, W, C, stride, kt):
    """small launches take two-row bands (dw2_fwd_kernel<T, 2>); the six-row form that every training layer uses is forced here"""
    be.lib.check(be.lib.fn["dev_set"](cabi.MDS_KNOB_DW2_R, 1), "dev_set")
    try:
        _dw_fwd_bwd(be, dt, N, T, H, W, C, stride, kt)
        test_dw_fwd_output_transform(be, dt, N, T, H, W, C, stride, kt)
    finally:
        be.lib.fn["dev_set"](cabi.MDS_KNOB_DW2_R, 0)


def _dw_fwd_bwd(be, dt, N, T, H, W, C, stride, kt):
    code, tdt = DT[dt]
    g = gen(H * W + C + kt)
    x = (torch.randn(N, C, T, H, W, generator=g) * 1.3).to(tdt)
    w = torch.randn(C, 1, kt, 3, 3, generator=g) * 0.3
    gamma = 1 + 0.2 * torch.randn(C, generator=g); beta = 0.2 * torch.randn(C, generator=g)
    eps = 1e-3
    OH, OW, pt, pl = geo.conv_geometry(H, W, stride)
    pads = (geo.same_pad(H, stride), geo.same_pad(W, stride)) if stride == 2 else ((1, 1), (1, 1))
    # reference: a = silu(bn(x)) with the batch statistics held constant; y = dw(a)
    xf = x.float()
    wf = w.clone().requires_grad_(True)
    mean = xf.mean((0, 2, 3, 4)); var = xf.var((0, 2, 3, 4), unbiased=False)
    rstd = 1 / torch.sqrt(var + eps)
    scale = gamma * rstd; shift = beta - mean * scale
    z = (xf * scale.view(1, -1, 1, 1, 1) + shift.view(1, -1, 1, 1, 1)).requires_grad_(True)
    a = F.silu(z)
    if dt == "bf16":
        a = a + (a.detach().to(tdt).float() - a.detach())               # bf16-rounded activations
    yref = dw_ref(a, wf, stride, kt, pads)
    dyt = torch.randn(yref.shape, generator=g).to(tdt)
    yref.backward(dyt.float())
    # forward kernel
    xd = be.t(to_rows(x)); wd = be.t(w.view(C, kt * 9))
    scd, shd = be.t(scale), be.t(shift)
    y = torch.full((N, T, OH, OW, C), float("nan")).to(tdt).to(be.device)
    st = torch.zeros(SLOTS, 2, C, device=be.device, dtype=torch.float64)
    pro = cabi.pro(2, scd, shd)
    be.call("dw_fwd", cabi.make("mds_dw_fwd_args", dtype=code, N=N, T=T, IH=H, IW=W, C=C, OH=OH, OW=OW, stride=stride,
                                pad_t=pt, pad_l=pl, kt=kt, x=xd, w=wd, y=y, pro=pro, stats=st))
    be.sync()
    yr = to_rows(yref.detach())
    assert_close(y, yr, dt, msg="y")
    s = st.sum(0).cpu()
    cnt = N * T * OH * OW
    assert_close(s[0], yr.sum((0, 1, 2, 3)), dt, scale=cnt ** 0.5, msg="sum")
    assert_close(s[1], (yr * yr).sum((0, 1, 2, 3)), dt, scale=cnt ** 0.5, msg="sumsq")
    # backward kernel
    gout = torch.full((N, T, H, W, C), float("nan")).to(tdt).to(be.device)
    dw = torch.zeros(C, kt * 9, device=be.device)
    st2 = torch.zeros(SLOTS, 2, C, device=be.device, dtype=torch.float64)     # backward sums: fp64 slots
    be.call("dw_bwd", cabi.make("mds_dw_bwd_args", dtype=code, N=N, T=T, IH=H, IW=W, C=C, OH=OH, OW=OW, stride=stride,
                                pad_t=pt, pad_l=pl, kt=kt, x=xd, dy=be.t(to_rows(dyt)), w=wd, g=gout, dw=dw, pro=pro,
                                mean=be.t(mean), rstd=be.t(rstd), stats=st2))
    be.sync()
    gref = to_rows(z.grad)
    assert_close(gout, gref, dt, msg="g")
    assert_close(dw, wf.grad.view(C, kt * 9), dt, scale=cnt ** 0.5, msg="dw")
    s2 = st2.sum(0).cpu()
    xhat = to_rows((xf - mean.view(1, -1, 1, 1, 1)) * rstd.view(1, -1, 1, 1, 1))
    cin = N * T * H * W
    assert_close(s2[0], gref.sum((0, 1, 2, 3)), dt, scale=cin ** 0.5, msg="sum g")
    assert_close(s2[1], (gref * xhat).sum((0, 1, 2, 3)), dt, scale=cin ** 0.5, msg="sum g*xhat")


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("N,T,H,W,C,stride,kt", [c for c in DW_CASES if c[6] == 1 or c[1] == 5])
def test_dw_fwd_output_transform(be, dt, N, T, H, W, C, stride, kt):
    """inference form: the input already is an activation (no prologue), the output is stored as silu(bn(y)) (mds_epi_t)"""
    code, tdt = DT[dt]
    g = gen(H * W + C + kt + 7)
    x = torch.randn(N, C, T, H, W, generator=g).to(tdt)
    w = torch.randn(C, 1, kt, 3, 3, generator=g) * 0.3
    esc = 1 + 0.3 * torch.randn(C, generator=g); esh = 0.4 * torch.randn(C, generator=g)
    OH, OW, pt, pl = geo.conv_geometry(H, W, stride)
    pads = (geo.same_pad(H, stride), geo.same_pad(W, stride)) if stride == 2 else ((1, 1), (1, 1))
    ref = F.silu(dw_ref(x.float(), w, stride, kt, pads) * esc.view(1, -1, 1, 1, 1) + esh.view(1, -1, 1, 1, 1))
    y = torch.full((N, T, OH, OW, C), float("nan")).to(tdt).to(be.device)
    # (kt == 1: T == 1 in every case; the pooled means of the STORED output come out of the same pass - mds_dw_fwd_args.pool)
    pool = torch.zeros(N, C, dtype=torch.float64, device=be.device)
    be.call("dw_fwd", cabi.make("mds_dw_fwd_args", dtype=code, N=N, T=T, IH=H, IW=W, C=C, OH=OH, OW=OW, stride=stride, pad_t=pt,
                                pad_l=pl, kt=kt, x=be.t(to_rows(x)), w=be.t(w.view(C, kt * 9)), y=y, pro=cabi.pro(0), stats=None,
                                epi=cabi.make("mds_epi_t", mode=2, scale=be.t(esc), shift=be.t(esh)),
                                pool=pool, pool_inv=1.0 / (T * OH * OW)))
    be.sync()
    assert_close(y, to_rows(ref), dt, msg="y")
    want = y.float().cpu().double().mean((1, 2, 3))
    assert (pool.cpu() - want).abs().max() <= 2e-6 * max(1.0, float(want.abs().max())), "pooled means of the stored output"


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("N,H,W", [(2, 20, 36), (1, 17, 23), (3, 6, 70), (2, 40, 150)])      # (the last: 3 x 3 tiles of the tiled kernels per image)
def test_stem_fwd_wgrad(be, dt, N, H, W):
    code, tdt = DT[dt]
    g = gen(H * W)
    x = torch.rand(N, 3, H, W, generator=g)
    w = torch.randn(32, 3, 3, 3, generator=g) * 0.3
    OH, OW, pt, pl = geo.conv_geometry(H, W, 2)
    (pt_, pb), (pl_, pr) = geo.same_pad(H, 2), geo.same_pad(W, 2)
    xq = x.to(tdt).float()
    wq = w.to(tdt).float().requires_grad_(True)
    yref = F.conv2d(F.pad(xq, (pl_, pr, pt_, pb)), wq, None, 2)
    dyt = torch.randn(yref.shape, generator=g).to(tdt)
    yref.backward(dyt.float())
    wp = torch.zeros(32, 32); wp[:, :27] = w.view(32, 27)
    y = torch.full((N, OH, OW, 32), float("nan")).to(tdt).to(be.device)
    st = torch.zeros(SLOTS, 2, 32, device=be.device, dtype=torch.float64)
    xd = be.t(x)
    be.call("stem_fwd", cabi.make("mds_stem_fwd_args", dtype=code, N=N, H=H, W=W, OH=OH, OW=OW, Cout=32, pad_t=pt, pad_l=pl,
                                  x=xd, w=be.t(wp.to(tdt)), y=y, stats=st))
    dw = torch.zeros(32, 3, 3, 3, device=be.device)
    be.call("stem_wgrad", cabi.make("mds_stem_wgrad_args", dtype=code, N=N, H=H, W=W, OH=OH, OW=OW, Cout=32, pad_t=pt,
                                    pad_l=pl, x=xd, dy=be.t(dyt.permute(0, 2, 3, 1)), dw=dw))
    be.sync()
    yr = yref.detach().permute(0, 2, 3, 1)
    assert_close(y, yr, dt, msg="y")
    s = st.sum(0).cpu()
    cnt = N * OH * OW
    assert_close(s[0], yr.sum((0, 1, 2)), dt, scale=cnt ** 0.5, msg="sum")
    assert_close(s[1], (yr * yr).sum((0, 1, 2)), dt, scale=cnt ** 0.5, msg="sumsq")
    assert_close(dw, wq.grad, dt, scale=cnt ** 0.5, msg="dw")


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_stem_fwd_output_transform(be, dt):
    code, tdt = DT[dt]
    N, H, W = 2, 18, 30
    g = gen(5)
    x = torch.rand(N, 3, H, W, generator=g)
    w = torch.randn(32, 3, 3, 3, generator=g) * 0.3
    esc = 1 + 0.3 * torch.randn(32, generator=g); esh = 0.4 * torch.randn(32, generator=g)
    OH, OW, pt, pl = geo.conv_geometry(H, W, 2)
    (pt_, pb), (pl_, pr) = geo.same_pad(H, 2), geo.same_pad(W, 2)
    ref = F.silu(F.conv2d(F.pad(x.to(tdt).float(), (pl_, pr, pt_, pb)), w.to(tdt).float(), None, 2).permute(0, 2, 3, 1) * esc + esh)
    wp = torch.zeros(32, 32); wp[:, :27] = w.view(32, 27)
    y = torch.full((N, OH, OW, 32), float("nan")).to(tdt).to(be.device)
    be.call("stem_fwd", cabi.make("mds_stem_fwd_args", dtype=code, N=N, H=H, W=W, OH=OH, OW=OW, Cout=32, pad_t=pt, pad_l=pl,
                                  x=be.t(x), w=be.t(wp.to(tdt)), y=y, stats=None,
                                  epi=cabi.make("mds_epi_t", mode=2, scale=be.t(esc), shift=be.t(esh))))
    be.sync()
    assert_close(y, ref, dt, msg="y")


@pytest.mark.parametrize("gmode", [cabi.MDS_G_PLAIN, cabi.MDS_G_SILU])
@pytest.mark.parametrize("N,H,W", [(2, 20, 36), (1, 17, 70)])
def test_stem_wgrad_forms_dy_on_load(be, gmode, N, H, W):
    """mds_stem_wgrad with a dy prologue (dy = A*g + B*y + D, g = u or u*silu'(y*scale + shift)) against the same kernel
    fed the materialised dy (what mds_bn_bwd_apply would have written)"""
    code, tdt = DT["bf16"]
    g = gen(H * W + gmode)
    OH, OW, pt, pl = geo.conv_geometry(H, W, 2)
    x = torch.rand(N, 3, H, W, generator=g)
    u = torch.randn(N, OH, OW, 32, generator=g).to(tdt)
    y = torch.randn(N, OH, OW, 32, generator=g).to(tdt)
    bn = torch.randn(4, 32, generator=g) * 0.5
    lin = torch.randn(3, 32, generator=g) * 0.5
    z = y.float() * bn[0] + bn[1]
    sg = torch.sigmoid(z)
    gg = u.float() * (sg * (1 + z * (1 - sg))) if gmode == cabi.MDS_G_SILU else u.float()
    dy = (lin[0] * gg + lin[1] * y.float() + lin[2]).to(tdt)
    xd = be.t(x)
    out = []
    for fused in (False, True):
        dw = torch.zeros(32, 3, 3, 3, device=be.device)
        kw = dict(dtype=code, N=N, H=H, W=W, OH=OH, OW=OW, Cout=32, pad_t=pt, pad_l=pl, x=xd, dw=dw)
        if fused:
            kw.update(dy=None, dyp=cabi.make("mds_dyp_t", mode=1, g=cabi.gsrc(gmode, be.t(u)), y=be.t(y), bn=be.t(bn), lin=be.t(lin)))
        else:
            kw.update(dy=be.t(dy))
        be.call("stem_wgrad", cabi.make("mds_stem_wgrad_args", **kw))
        be.sync()
        out.append(dw.cpu())
    # the fused path multiplies unrounded fp32 dy, the other one bf16-rounded dy: bf16-level agreement
    assert_close(out[1], out[0], "bf16", scale=(N * OH * OW) ** 0.5, msg="dw")


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_stem_fwd_uint8_ingest(be, dt):
    """raw uint8 frames -> constant pad (src/frames.py:12-31) -> /255 -> kornia hflip for the TTA copies, all
    inside the stem's gather, against the reference's pipeline followed by the convolution."""
    code, tdt = DT[dt]
    g = gen(77)
    nsrc, sh, sw, H, W = 2, 21, 38, 26, 44                      # padded size larger than the frames in both directions
    u8 = torch.randint(0, 256, (nsrc, 3, sh, sw), generator=g, dtype=torch.uint8)
    w = torch.randn(32, 3, 3, 3, generator=g) * 0.3
    hp, wp_ = H - sh, W - sw
    fr = F.pad(u8, [wp_ // 2, wp_ - wp_ // 2, hp // 2, hp - hp // 2], mode="constant", value=0).to(torch.float32) / 255.0
    x = torch.cat([fr, torch.flip(fr, dims=[-1])], dim=0)         # images 0..nsrc-1, then their mirrored copies
    N = 2 * nsrc
    OH, OW, pt, pl = geo.conv_geometry(H, W, 2)
    (pt_, pb), (pl_, pr) = geo.same_pad(H, 2), geo.same_pad(W, 2)
    yref = F.conv2d(F.pad(x.to(tdt).float(), (pl_, pr, pt_, pb)), w.to(tdt).float(), None, 2).permute(0, 2, 3, 1)
    wp = torch.zeros(32, 32); wp[:, :27] = w.view(32, 27)
    y = torch.full((N, OH, OW, 32), float("nan")).to(tdt).to(be.device)
    ing = cabi.make("mds_ingest_t", u8=be.t(u8), nsrc=nsrc, src_h=sh, src_w=sw, pad_top=hp // 2, pad_left=wp_ // 2, scale=1.0 / 255.0)
    be.call("stem_fwd", cabi.make("mds_stem_fwd_args", dtype=code, N=N, H=H, W=W, OH=OH, OW=OW, Cout=32, pad_t=pt, pad_l=pl,
                                  x=None, w=be.t(wp.to(tdt)), y=y, stats=None, ingest=ing))
    be.sync()
    assert_close(y, yref, dt, msg="y")
