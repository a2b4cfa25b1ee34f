"""Dense 3x3 conv MFMA kernels (fwd, both dgrads, wgrad) against torch conv2d / autograd."""
import pytest
import torch
import torch.nn.functional as F

from backends import be, be_gpu, DT, assert_close  # noqa: F401
from mds import cabi, geometry as geo
from oracle.multidim_stacker_ref import Conv2dSame


def gen(s):
    return torch.Generator().manual_seed(s)


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def ref_conv(x, w, stride):
    if stride == 1:
        return F.conv2d(x, w, None, 1, 1)
    (pt, pb), (pl, pr) = geo.same_pad(x.shape[2], stride), geo.same_pad(x.shape[3], stride)
    ref_mod = Conv2dSame(w.shape[1], w.shape[0], 3, stride, bias=False)   # the oracle's padding rule
    with torch.no_grad():
        probe = torch.zeros(1, w.shape[1], x.shape[2], x.shape[3])
        assert ref_mod(probe).shape[2:] == F.conv2d(F.pad(probe, (pl, pr, pt, pb)), w.detach(), None, stride).shape[2:]
    return F.conv2d(F.pad(x, (pl, pr, pt, pb)), w, None, stride)


def pack(w, kind, tdt):
    O, I = w.shape[:2]
    w9 = w.reshape(O, I, 9)
    if kind == "oi":
        return w9.permute(0, 2, 1).contiguous().to(tdt)          # [O][9][I]
    return w9.flip(2).permute(1, 2, 0).contiguous().to(tdt)      # [I][9 flipped][O]


CASES = [  # N, H, W, Cin, Cout, stride, pro
    (2, 13, 21, 32, 16, 1, 2),
    (1, 9, 18, 48, 192, 1, 0),
    (2, 16, 34, 16, 64, 2, 2),
    (1, 11, 17, 32, 128, 2, 1),   # odd sizes: TF-SAME pads 1/1
]


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("N,H,W,Cin,Cout,stride,mode", CASES)
def test_conv_fwd(be, dt, N, H, W, Cin, Cout, stride, mode):
    code, tdt = DT[dt]
    g = gen(H * W + Cin)
    x = torch.randn(N, Cin, H, W, generator=g).to(tdt)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5).to(tdt)
    scale = 1 + 0.2 * torch.randn(Cin, generator=g); shift = 0.3 * torch.randn(Cin, generator=g)
    OH, OW, pt, pl = geo.conv_geometry(H, W, stride)
    dy, dx, wi = geo.taps_fwd(pt, pl)
    y = torch.full((N, OH, OW, Cout), float("nan")).to(tdt).to(be.device)
    st = torch.zeros(cabi.MDS_STAT_SLOTS, 2, Cout, device=be.device, dtype=torch.float64)
    args = cabi.make("mds_conv_fwd_args", dtype=code, N=N, IH=H, IW=W, Cin=Cin, OH=OH, OW=OW, Cout=Cout,
                     A=OH, B=OW, oy0=0, ox0=0, os=1, **{"is": stride}, ntaps=9, dy=dy, dx=dx, wi=wi, wtaps=9,
                     x=be.t(nhwc(x)), w=be.t(pack(w, "oi", tdt)), y=y,
                     pro=cabi.pro(mode, be.t(scale), be.t(shift)), residual=None, stats=st)
    be.call("conv_fwd", args)
    be.sync()
    a = x.float()
    if mode:
        a = a * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
        if mode == 2:
            a = F.silu(a)
        if dt == "bf16":
            a = a.to(tdt).float()
    ref = nhwc(ref_conv(a, w.float(), stride))
    assert ref.shape == y.shape
    assert_close(y, ref, dt, msg="y")
    s = st.sum(0).cpu()
    cnt = N * OH * OW
    assert_close(s[0], ref.sum((0, 1, 2)), dt, scale=cnt ** 0.5, msg="sum")
    assert_close(s[1], (ref * ref).sum((0, 1, 2)), dt, scale=cnt ** 0.5, msg="sumsq")


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("N,H,W,Cin,Cout,stride,mode", CASES)
def test_conv_dgrad(be, dt, N, H, W, Cin, Cout, stride, mode):
    code, tdt = DT[dt]
    g = gen(H + W + Cout)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5).to(tdt)
    OH, OW, pt, pl = geo.conv_geometry(H, W, stride)
    dyt = torch.randn(N, Cout, OH, OW, generator=g).to(tdt)
    res = torch.randn(N, H, W, Cin, generator=g).to(tdt)
    xx = torch.zeros(N, Cin, H, W, requires_grad=True)
    ref_conv(xx, w.float(), stride).backward(dyt.float())
    ref = nhwc(xx.grad) + res.float()
    dxo = torch.full((N, H, W, Cin), float("nan")).to(tdt).to(be.device)
    wp = be.t(pack(w, "io", tdt)); dyd = be.t(nhwc(dyt)); rd = be.t(res)
    common = dict(dtype=code, N=N, IH=OH, IW=OW, Cin=Cout, OH=H, OW=W, Cout=Cin, wtaps=9, x=dyd, w=wp, y=dxo,
                  pro=cabi.pro(0), residual=rd, stats=None)
    if stride == 1:
        dy, dx, wi = geo.taps_dgrad_s1()
        be.call("conv_fwd", cabi.make("mds_conv_fwd_args", A=H, B=W, oy0=0, ox0=0, os=1, **{"is": 1}, ntaps=9,
                                      dy=dy, dx=dx, wi=wi, **common))
    else:
        for py in range(2):
            for px in range(2):
                dy, dx, wi = geo.taps_dgrad_s2(py, px, pt, pl)
                A, B = (H - py + 1) // 2, (W - px + 1) // 2
                if A <= 0 or B <= 0 or not dy:
                    continue
                be.call("conv_fwd", cabi.make("mds_conv_fwd_args", A=A, B=B, oy0=py, ox0=px, os=2, **{"is": 1},
                                              ntaps=len(dy), dy=dy, dx=dx, wi=wi, **common))
    be.sync()
    assert_close(dxo, ref, dt, msg="dx")


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("N,H,W,Cin,Cout,stride,emode,res", [(2, 13, 21, 32, 16, 1, 2, False), (1, 16, 34, 16, 64, 2, 2, False),
                                                             (1, 9, 18, 48, 192, 1, 1, True), (1, 12, 20, 128, 32, 1, 2, True)])
def test_conv_fwd_output_transform(be, dt, N, H, W, Cin, Cout, stride, emode, res):
    """mds_epi_t in both convolution kernels: y = act(acc*scale + shift) (+ residual); no prologue (activated input)"""
    code, tdt = DT[dt]
    g = gen(H * W + Cin + emode)
    x = torch.randn(N, Cin, H, W, generator=g).to(tdt)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5).to(tdt)
    esc = 1 + 0.3 * torch.randn(Cout, generator=g); esh = 0.4 * torch.randn(Cout, generator=g)
    OH, OW, pt, pl = geo.conv_geometry(H, W, stride)
    dy, dx, wi = geo.taps_fwd(pt, pl)
    r = torch.randn(N, OH, OW, Cout, generator=g).to(tdt) if res else None
    y = torch.full((N, OH, OW, Cout), float("nan")).to(tdt).to(be.device)
    be.call("conv_fwd", cabi.make("mds_conv_fwd_args", dtype=code, N=N, IH=H, IW=W, Cin=Cin, OH=OH, OW=OW, Cout=Cout, A=OH, B=OW,
                                  oy0=0, ox0=0, os=1, **{"is": stride}, ntaps=9, dy=dy, dx=dx, wi=wi, wtaps=9, x=be.t(nhwc(x)),
                                  w=be.t(pack(w, "oi", tdt)), y=y, pro=cabi.pro(0), residual=be.t(r) if res else None, stats=None,
                                  epi=cabi.make("mds_epi_t", mode=emode, scale=be.t(esc), shift=be.t(esh))))
    be.sync()
    ref = nhwc(ref_conv(x.float(), w.float(), stride)) * esc + esh
    if emode == 2:
        ref = F.silu(ref)
    if res:
        ref = ref + r.float()
    assert_close(y, ref, dt, msg="y")


GROUP_CASES = [  # N, H, W, Cin, Cout (of the forward stride-2 conv), persistent blocks (0 = default grid)
    (2, 16, 34, 16, 64, 0),
    (1, 11, 17, 32, 128, 0),      # odd sizes: the parities' sub-grids differ in extent
    (2, 40, 70, 16, 64, 3),       # 3 blocks: each walks several tiles through both register sets (+ an odd tail)
]


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("N,H,W,Cin,Cout,blocks", GROUP_CASES)
def test_conv_dgrad_s2_tap_groups(be, dt, N, H, W, Cin, Cout, blocks):
    """the four output parities of the stride-2 data gradient as tap groups of ONE launch (engine._conv_dgrad)"""
    code, tdt = DT[dt]
    g = gen(H + W + Cout + 1)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5).to(tdt)
    OH, OW, pt, pl = geo.conv_geometry(H, W, 2)
    dyt = torch.randn(N, Cout, OH, OW, generator=g).to(tdt)
    xx = torch.zeros(N, Cin, H, W, requires_grad=True)
    ref_conv(xx, w.float(), 2).backward(dyt.float())
    ref = nhwc(xx.grad)
    dxo = torch.full((N, H, W, Cin), float("nan")).to(tdt).to(be.device)
    par = []
    for py in range(2):
        for px in range(2):
            dy, dx, wi = geo.taps_dgrad_s2(py, px, pt, pl)
            par.append((py, px, dy, dx, wi, (H - py + 1) // 2, (W - px + 1) // 2))
    args = cabi.make("mds_conv_fwd_args", dtype=code, N=N, IH=OH, IW=OW, Cin=Cout, OH=H, OW=W, Cout=Cin, wtaps=9,
                     x=be.t(nhwc(dyt)), w=be.t(pack(w, "io", tdt)), y=dxo, pro=cabi.pro(0), residual=None, stats=None,
                     A=max(p[5] for p in par), B=max(p[6] for p in par), oy0=0, ox0=0, os=2, **{"is": 1},
                     ntaps=sum(len(p[2]) for p in par), dy=sum((p[2] for p in par), []), dx=sum((p[3] for p in par), []),
                     wi=sum((p[4] for p in par), []), ngroups=4, g_ntaps=[len(p[2]) for p in par],
                     g_oy0=[p[0] for p in par], g_ox0=[p[1] for p in par], g_A=[p[5] for p in par], g_B=[p[6] for p in par])
    be.lib.check(be.lib.fn["dev_set"](cabi.MDS_KNOB_CONV_BLOCKS, blocks), "dev_set")
    try:
        be.call("conv_fwd", args)
        be.sync()
    finally:
        be.lib.fn["dev_set"](cabi.MDS_KNOB_CONV_BLOCKS, 0)
    assert_close(dxo, ref, dt, msg="dx")


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("N,H,W,Cin,Cout,stride,mode,blocks", [(2, 45, 50, 32, 16, 1, 2, 5), (1, 40, 70, 16, 64, 2, 2, 2),
                                                               (1, 33, 40, 32, 128, 1, 0, 3), (1, 30, 37, 16, 32, 1, 1, 4)])
def test_conv_fwd_persistent_pipeline(be, dt, N, H, W, Cin, Cout, stride, mode, blocks):
    """few persistent blocks: every block walks many tiles through the two prefetch register sets (even / odd counts)"""
    be.lib.check(be.lib.fn["dev_set"](cabi.MDS_KNOB_CONV_BLOCKS, blocks), "dev_set")
    try:
        test_conv_fwd(be, dt, N, H, W, Cin, Cout, stride, mode)
    finally:
        be.lib.fn["dev_set"](cabi.MDS_KNOB_CONV_BLOCKS, 0)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("N,H,W,Cin,Cout,stride,mode", CASES)
def test_conv_wgrad(be, dt, N, H, W, Cin, Cout, stride, mode):
    code, tdt = DT[dt]
    g = gen(H * 3 + W + Cout)
    x = torch.randn(N, Cin, H, W, generator=g).to(tdt)
    scale = 1 + 0.2 * torch.randn(Cin, generator=g); shift = 0.3 * torch.randn(Cin, generator=g)
    OH, OW, pt, pl = geo.conv_geometry(H, W, stride)
    dyt = torch.randn(N, Cout, OH, OW, generator=g).to(tdt)
    a = x.float()
    if mode:
        a = a * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
        if mode == 2:
            a = F.silu(a)
        if dt == "bf16":
            a = a.to(tdt).float()
    ww = torch.zeros(Cout, Cin, 3, 3, requires_grad=True)
    ref_conv(a, ww, stride).backward(dyt.float())
    dy, dx, wi = geo.taps_fwd(pt, pl)
    dw = torch.zeros(Cout, Cin, 3, 3, device=be.device)
    be.call("conv_wgrad", cabi.make("mds_conv_wgrad_args", dtype=code, N=N, IH=H, IW=W, Cin=Cin, OH=OH, OW=OW,
                                    Cout=Cout, **{"is": stride}, ntaps=9, dy=dy, dx=dx, wi=wi, wtaps=9,
                                    x=be.t(nhwc(x)), dyt=be.t(nhwc(dyt)), dw=dw,
                                    pro=cabi.pro(mode, be.t(scale), be.t(shift))))
    be.sync()
    assert_close(dw, ww.grad, dt, scale=(N * OH * OW) ** 0.5, msg="dw")


C3_CASES = [  # N, H, W, Cin, Cout, residual, statistics, block cap (0 = default grid)
    (2, 13, 37, 32, 128, False, True, 0),      # forward shape of blocks.1.1 / 2.0's stride-1 sibling: ragged band, two images
    (1, 23, 64, 32, 128, False, True, 3),      # three blocks walk several items each: the DMA ring runs across item boundaries
    (2, 11, 21, 128, 32, True, False, 0),      # its data gradient: residual operand through the ring
    (1, 40, 70, 128, 32, True, False, 2),
    (1, 9, 33, 128, 32, False, False, 1),      # one block, every item; no residual
    (1, 3, 16, 32, 128, False, True, 0),       # fewer rows than the ring is deep
    (1, 2, 15, 128, 32, True, False, 0),
    (2, 12, 37, 48, 192, False, True, 0),      # blocks.2.1's forward: three channel passes of 64
    (1, 20, 50, 48, 192, False, True, 2),
    (2, 9, 70, 16, 32, False, False, 0),       # blocks.0.0's data gradient: 16 input channels (half-empty second k-step), 64-column bands
    (1, 17, 130, 16, 32, True, False, 3),
]


@pytest.mark.parametrize("N,H,W,Cin,Cout,res,stats,blocks", C3_CASES)
def test_c3_filter_in_registers(be, N, H, W, Cin, Cout, res, stats, blocks):
    """k_c3.hip (bf16, stride 1, no prologue): producer / consumer row ring, rolling output rows, rotated LDS parts, zero-page
    borders - against torch's conv2d; MDS_KNOB_C3 = 2 sends every legal shape there whatever its size"""
    code, tdt = DT["bf16"]
    g = gen(H * W + Cin + Cout)
    x = torch.randn(N, Cin, H, W, generator=g).to(tdt)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5).to(tdt)
    r = torch.randn(N, H, W, Cout, generator=g).to(tdt) if res else None
    dy, dx, wi = geo.taps_fwd(1, 1)
    y = torch.full((N, H, W, Cout), float("nan")).to(tdt).to(be.device)
    st = torch.zeros(cabi.MDS_STAT_SLOTS, 2, Cout, device=be.device, dtype=torch.float64) if stats else None
    args = cabi.make("mds_conv_fwd_args", dtype=code, N=N, IH=H, IW=W, Cin=Cin, OH=H, OW=W, Cout=Cout, A=H, B=W, oy0=0, ox0=0, os=1,
                     **{"is": 1}, ntaps=9, dy=dy, dx=dx, wi=wi, wtaps=9, x=be.t(nhwc(x)), w=be.t(pack(w, "oi", tdt)), y=y,
                     pro=cabi.pro(0), residual=be.t(r) if res else None, stats=st)
    be.lib.check(be.lib.fn["dev_set"](cabi.MDS_KNOB_C3, 2), "dev_set")
    be.lib.check(be.lib.fn["dev_set"](cabi.MDS_KNOB_CONV_BLOCKS, blocks), "dev_set")
    try:
        be.call("conv_fwd", args)
        be.sync()
    finally:
        be.lib.fn["dev_set"](cabi.MDS_KNOB_C3, 0)
        be.lib.fn["dev_set"](cabi.MDS_KNOB_CONV_BLOCKS, 0)
    ref = nhwc(ref_conv(x.float(), w.float(), 1))
    if stats:
        s = st.sum(0).cpu()
        assert_close(s[0], ref.sum((0, 1, 2)), "bf16", scale=(N * H * W) ** 0.5, msg="sum")
        assert_close(s[1], (ref * ref).sum((0, 1, 2)), "bf16", scale=(N * H * W) ** 0.5, msg="sumsq")
    if res:
        ref = ref + r.float()
    assert_close(y, ref, "bf16", msg="y")


C3T_CASES = [  # N, H, W (of the layer's input = the gradient that comes out), Cin, Cout of the FORWARD layer, block cap
    (2, 12, 64, 32, 128, 0),       # blocks.2.0's data gradient (128 -> 32 channels): two images, exact bands
    (1, 22, 40, 32, 128, 2),       # ragged band (20 gradient columns per 32-column band), two blocks walk several items
    (1, 8, 128, 16, 64, 0),        # blocks.1.0's (64 -> 16): 64-column bands, one wave per strip
    (1, 30, 70, 16, 64, 3),
    (1, 2, 32, 32, 128, 1),        # a single input row per item
]


@pytest.mark.parametrize("N,H,W,Cin,Cout,blocks", C3T_CASES)
def test_c3t_stride2_data_gradient(be, N, H, W, Cin, Cout, blocks):
    """k_c3.hip's transposed-convolution kernel (bf16, even extents: TF-SAME pads 0 / 1) through the tap-group form of
    mds_conv_fwd that engine._conv_dgrad launches, against autograd; MDS_KNOB_C3 = 2 lifts the size bar"""
    be.lib.check(be.lib.fn["dev_set"](cabi.MDS_KNOB_C3, 2), "dev_set")
    try:
        test_conv_dgrad_s2_tap_groups(be, "bf16", N, H, W, Cin, Cout, blocks)
    finally:
        be.lib.fn["dev_set"](cabi.MDS_KNOB_C3, 0)


@pytest.mark.parametrize("stride,N,H,W,blocks,masked,Cin,Cout", [(1, 2, 14, 64, 0, True, 32, 128), (1, 1, 21, 45, 2, False, 32, 128), (2, 2, 12, 64, 0, True, 32, 128),
                                                                  (2, 1, 22, 40, 2, False, 32, 128), (1, 2, 9, 128, 0, "silu", 32, 16), (1, 1, 20, 70, 2, "silu", 32, 16),
                                                                  (2, 1, 10, 128, 0, "silu", 16, 64), (2, 2, 6, 64, 2, "silu", 32, 128)])
def test_c3_post_statistics(be, stride, N, H, W, blocks, masked, Cin, Cout):
    """mds_poststat_t in k_c3.hip's data gradients (what engine._conv_dgrad asks for when mds_conv_dgrad_post_ok says 1): the
    output u is the gradient source of the BatchNorm below - sum g and sum g * xhat over all pixels with g = bf16(u) (x DropPath's
    per-image factor), next to the unchanged u; stride 1 = blocks.1.1's form (32 -> 128 forward, residual operand), stride 2 =
    blocks.2.0's"""
    code, tdt = DT["bf16"]                    # (Cin, Cout: the forward layer - the gradient that comes out has Cin channels)
    silu, masked = masked == "silu", masked is True
    g = gen(H * W + stride)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5).to(tdt)
    OH, OW, pt, pl = geo.conv_geometry(H, W, stride)
    dyt = torch.randn(N, Cout, OH, OW, generator=g).to(tdt)
    res = torch.randn(N, H, W, Cin, generator=g).to(tdt) if (stride == 1 and Cout == 128) else None
    yb = torch.randn(N, H, W, Cin, generator=g).to(tdt)
    bn = torch.stack([1 + 0.2 * torch.randn(Cin, generator=g), 0.3 * torch.randn(Cin, generator=g), 0.3 * torch.randn(Cin, generator=g), 0.5 + torch.rand(Cin, generator=g)])
    mask = torch.tensor([1.25, 0.0][:N]) if masked else None
    assert be.lib.fn["conv_dgrad_post_ok"](code, N, H, W, Cin, Cout, stride, int(res is not None)) == 0     # small: below the size bar
    be.lib.check(be.lib.fn["dev_set"](cabi.MDS_KNOB_C3, 2), "dev_set")
    be.lib.check(be.lib.fn["dev_set"](cabi.MDS_KNOB_CONV_BLOCKS, blocks), "dev_set")
    try:
        assert be.lib.fn["conv_dgrad_post_ok"](code, N, H, W, Cin, Cout, stride, int(res is not None)) == 1
        xx = torch.zeros(N, Cin, H, W, requires_grad=True)
        ref_conv(xx, w.float(), stride).backward(dyt.float())
        ref = nhwc(xx.grad) + (res.float() if res is not None else 0)
        dxo = torch.full((N, H, W, Cin), float("nan")).to(tdt).to(be.device)
        st = torch.zeros(cabi.MDS_STAT_SLOTS, 2, Cin, device=be.device, dtype=torch.float64)
        post = cabi.make("mds_poststat_t", mode=3 if silu else (2 if masked else 1), y=be.t(yb), bn=be.t(bn), mask=be.t(mask) if masked else None,
                         rows_per_group=H * W, stats=st)
        common = dict(dtype=code, N=N, IH=OH, IW=OW, Cin=Cout, OH=H, OW=W, Cout=Cin, wtaps=9, x=be.t(nhwc(dyt)), w=be.t(pack(w, "io", tdt)),
                      y=dxo, pro=cabi.pro(0), residual=be.t(res) if res is not None else None, stats=None, post=post)
        if stride == 1:
            dy, dx, wi = geo.taps_dgrad_s1()
            args = cabi.make("mds_conv_fwd_args", A=H, B=W, oy0=0, ox0=0, os=1, **{"is": 1}, ntaps=9, dy=dy, dx=dx, wi=wi, **common)
        else:
            par = []
            for py in range(2):
                for px in range(2):
                    dy, dx, wi = geo.taps_dgrad_s2(py, px, pt, pl)
                    par.append((py, px, dy, dx, wi, (H - py + 1) // 2, (W - px + 1) // 2))
            args = cabi.make("mds_conv_fwd_args", A=max(p[5] for p in par), B=max(p[6] for p in par), oy0=0, ox0=0, os=2, **{"is": 1},
                             ntaps=sum(len(p[2]) for p in par), dy=sum((p[2] for p in par), []), dx=sum((p[3] for p in par), []),
                             wi=sum((p[4] for p in par), []), ngroups=4, g_ntaps=[len(p[2]) for p in par],
                             g_oy0=[p[0] for p in par], g_ox0=[p[1] for p in par], g_A=[p[5] for p in par], g_B=[p[6] for p in par], **common)
        be.call("conv_fwd", args)
        be.sync()
    finally:
        be.lib.fn["dev_set"](cabi.MDS_KNOB_C3, 0)
        be.lib.fn["dev_set"](cabi.MDS_KNOB_CONV_BLOCKS, 0)
    if silu:      # MDS_POST_SILU: g = u * silu'(y * scale + shift) replaces u in memory
        z = yb.float() * bn[0] + bn[1]
        sg = torch.sigmoid(z)
        ref = ref.to(tdt).float() * (sg * (1 + z * (1 - sg)))
    assert_close(dxo, ref, "bf16", msg="dx")
    u = dxo.float().cpu()                                   # the sums are taken of what later readers will read
    gg = u * (mask.view(N, 1, 1, 1) if masked else 1.0)
    xhat = (yb.float() - bn[2]) * bn[3]
    s = st.sum(0).cpu()
    cnt = N * H * W
    assert_close(s[0], gg.sum((0, 1, 2)), "bf16", scale=cnt ** 0.5, msg="sum g")
    assert_close(s[1], (gg * xhat).sum((0, 1, 2)), "bf16", scale=cnt ** 0.5, msg="sum g xhat")
    # the old kernels refuse post statistics instead of ignoring them
    be.lib.check(be.lib.fn["dev_set"](cabi.MDS_KNOB_C3, 1), "dev_set")
    try:
        with pytest.raises(Exception):
            be.call("conv_fwd", args)
    finally:
        be.lib.fn["dev_set"](cabi.MDS_KNOB_C3, 0)


@pytest.mark.parametrize("N,H,W,blocks", [(2, 13, 128, 0), (1, 21, 70, 2), (1, 40, 64, 1), (1, 2, 64, 0)])
def test_c3_prologue_through_transform_waves(be, N, H, W, blocks):
    """the first 3x3 layer (32 -> 16, input = the stem's raw output read through BatchNorm + SiLU): k_c3.hip's transform waves rewrite
    the landed rows in place one batch ahead of the consumers; zero padding is applied AFTER the activation"""
    be.lib.check(be.lib.fn["dev_set"](cabi.MDS_KNOB_C3, 2), "dev_set")
    be.lib.check(be.lib.fn["dev_set"](cabi.MDS_KNOB_CONV_BLOCKS, blocks), "dev_set")
    try:
        test_conv_fwd(be, "bf16", N, H, W, 32, 16, 1, 2)
    finally:
        be.lib.fn["dev_set"](cabi.MDS_KNOB_C3, 0)
        be.lib.fn["dev_set"](cabi.MDS_KNOB_CONV_BLOCKS, 0)


C3S_CASES = [  # N, H, W (even: TF-SAME pads 0 / 1), Cin, Cout, prologue mode, block cap
    (2, 12, 128, 32, 128, 0, 0),      # blocks.2.0's first convolution: two exact 32-column output bands, two images
    (1, 22, 40, 32, 128, 0, 2),       # ragged band (20 output columns), an odd number of output rows, two blocks walk several items
    (1, 8, 256, 16, 64, 0, 0),        # blocks.1.0's shape without the prologue: 64-column output bands, half-empty second k-step
    (1, 30, 70, 16, 64, 0, 3),
    (1, 2, 32, 32, 128, 0, 1),        # a single output row: one even input row + the pad row
    (2, 10, 128, 16, 64, 2, 0),       # blocks.1.0: the input is blocks.0.0's raw output read through BatchNorm + SiLU (transform waves)
    (1, 26, 256, 16, 64, 2, 2),       # ... two blocks walk several items; the pad column (W) and row (H) are zero AFTER the activation
    (1, 4, 128, 16, 64, 2, 1),
    (1, 26, 100, 16, 64, 2, 2),       # a ragged width behind the prologue is refused (k_conv.hip takes it): same answer
]


@pytest.mark.parametrize("N,H,W,Cin,Cout,mode,blocks", C3S_CASES)
def test_c3s_stride2_forward(be, N, H, W, Cin, Cout, mode, blocks):
    """k_c3.hip's stride-2 forward kernel (bf16, even extents): two rolling accumulator sets over batches of four input rows,
    pixel-stride-2 fragment reads, zero-page pad row / column, forward statistics - against torch's conv2d through test_conv_fwd"""
    be.lib.check(be.lib.fn["dev_set"](cabi.MDS_KNOB_C3, 2), "dev_set")
    be.lib.check(be.lib.fn["dev_set"](cabi.MDS_KNOB_CONV_BLOCKS, blocks), "dev_set")
    try:
        test_conv_fwd(be, "bf16", N, H, W, Cin, Cout, 2, mode)
    finally:
        be.lib.fn["dev_set"](cabi.MDS_KNOB_C3, 0)
        be.lib.fn["dev_set"](cabi.MDS_KNOB_CONV_BLOCKS, 0)


C3W_CASES = [  # N, H, W, Cin, Cout, block cap
    (2, 13, 37, 32, 128, 0),       # blocks.1.1's weight gradient: ragged 32-column band, two images
    (1, 23, 64, 32, 128, 3),       # three blocks walk several items each: the ring runs across item boundaries
    (2, 12, 37, 48, 192, 0),       # blocks.2.1's: three channel passes of 64
    (1, 20, 50, 48, 192, 2),
    (1, 3, 32, 32, 128, 1),        # fewer rows than the ring is deep
    (1, 1, 16, 48, 192, 0),        # a single image row: one input row between two zero entries
]


@pytest.mark.parametrize("N,H,W,Cin,Cout,blocks", C3W_CASES)
def test_c3w_weight_gradient_row_streaming(be, N, H, W, Cin, Cout, blocks):
    """k_c3.hip's weight-gradient kernel (bf16, stride 1, no prologue): pixel-major rows of input and dy through the DMA ring, both
    MFMA operands by transposing LDS reads, dy zero outside an item's rows, OIHW flush through LDS - against autograd"""
    be.lib.check(be.lib.fn["dev_set"](cabi.MDS_KNOB_C3, 2), "dev_set")
    be.lib.check(be.lib.fn["dev_set"](cabi.MDS_KNOB_CONV_BLOCKS, blocks), "dev_set")
    try:
        test_conv_wgrad(be, "bf16", N, H, W, Cin, Cout, 1, 0)
    finally:
        be.lib.fn["dev_set"](cabi.MDS_KNOB_C3, 0)
        be.lib.fn["dev_set"](cabi.MDS_KNOB_CONV_BLOCKS, 0)


@pytest.mark.parametrize("N,H,W,blocks", [(2, 9, 128, 0), (1, 20, 70, 2), (1, 3, 64, 1), (1, 14, 200, 3)])
def test_c3w_weight_gradient_behind_the_prologue(be, N, H, W, blocks):
    """blocks.0.0's weight gradient (32 -> 16, input = the stem's raw output through BatchNorm + SiLU): transform waves on the input part
    of the ring entries, consumers one barrier behind, pixel-split consumer waves summed through LDS; zero padding after the activation"""
    be.lib.check(be.lib.fn["dev_set"](cabi.MDS_KNOB_C3, 2), "dev_set")
    be.lib.check(be.lib.fn["dev_set"](cabi.MDS_KNOB_CONV_BLOCKS, blocks), "dev_set")
    try:
        test_conv_wgrad(be, "bf16", N, H, W, 32, 16, 1, 2)
    finally:
        be.lib.fn["dev_set"](cabi.MDS_KNOB_C3, 0)
        be.lib.fn["dev_set"](cabi.MDS_KNOB_CONV_BLOCKS, 0)


@pytest.mark.parametrize("N,H,W,blocks", [(2, 12, 128, 0), (1, 22, 40, 2), (1, 2, 64, 1), (1, 30, 200, 3)])
def test_c3w2_stride2_weight_gradient(be, N, H, W, blocks):
    """blocks.2.0's weight gradient (32 -> 128, stride 2, TF-SAME pads 0 / 1): a ring entry = a dy row with its two input rows, the even
    row also meets the dy row above (ky = 2), pixel-stride-2 transposing reads, one closing entry per item - against autograd"""
    be.lib.check(be.lib.fn["dev_set"](cabi.MDS_KNOB_C3, 2), "dev_set")
    be.lib.check(be.lib.fn["dev_set"](cabi.MDS_KNOB_CONV_BLOCKS, blocks), "dev_set")
    try:
        test_conv_wgrad(be, "bf16", N, H, W, 32, 128, 2, 0)
    finally:
        be.lib.fn["dev_set"](cabi.MDS_KNOB_C3, 0)
        be.lib.fn["dev_set"](cabi.MDS_KNOB_CONV_BLOCKS, 0)


TRAIN_LAYERS = [  # the 3x3 layers of the benchmarked step (20 images): H, W, Cin, Cout, stride, prologue
    (368, 640, 32, 16, 1, 2),       # blocks.0.0 (behind the stem's BatchNorm + SiLU)
    (368, 640, 16, 64, 2, 2),       # blocks.1.0 first convolution (behind blocks.0.0's)
    (184, 320, 32, 128, 1, 0),      # blocks.1.1
    (184, 320, 32, 128, 2, 0),      # blocks.2.0
    (92, 160, 48, 192, 1, 0),       # blocks.2.1
]


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,Cin,Cout,stride,mode", TRAIN_LAYERS)
def test_k_c3_at_the_training_sizes_against_k_conv(be_gpu, H, W, Cin, Cout, stride, mode):
    """BASELINE config 2's own layer sizes (the sizes the row-streaming kernels are dispatched at without a knob): forward output +
    forward statistics and the weight gradient from k_c3.hip against k_conv.hip's kernels (MDS_KNOB_C3 = 1) on the same tensors.
    Both are bf16 MFMA paths with fp32 accumulation: outputs agree to bf16 rounding, sums to their fp32 partial-sum order."""
    import os
    be, N = be_gpu, int(os.environ.get("MDS_TEST_IMAGES", "20"))      # (20 = batch 4; 5 / 10 / 15 - other item and block counts - were run by hand: profiles/r06_other_batch_sizes.txt)
    code, tdt = DT["bf16"]
    g = torch.Generator(device=be.device).manual_seed(H + Cin + stride)
    x = torch.randn(N, H, W, Cin, device=be.device, generator=g).to(tdt)
    w = (torch.randn(Cout, Cin, 3, 3, device=be.device, generator=g) / (9 * Cin) ** 0.5).to(tdt)
    scale = 1 + 0.2 * torch.randn(Cin, device=be.device, generator=g); shift = 0.3 * torch.randn(Cin, device=be.device, generator=g)
    OH, OW, pt, pl = geo.conv_geometry(H, W, stride)
    dyt = torch.randn(N, OH, OW, Cout, device=be.device, generator=g).to(tdt)
    dy, dx, wi = geo.taps_fwd(pt, pl)
    wp = be.t(pack(w.cpu(), "oi", tdt))
    out = {}
    for knob in (0, 1):
        be.lib.check(be.lib.fn["dev_set"](cabi.MDS_KNOB_C3, knob), "dev_set")
        try:
            y = torch.full((N, OH, OW, Cout), float("nan"), device=be.device).to(tdt)
            st = torch.zeros(cabi.MDS_STAT_SLOTS, 2, Cout, device=be.device, dtype=torch.float64)
            be.call("conv_fwd", cabi.make("mds_conv_fwd_args", dtype=code, N=N, IH=H, IW=W, Cin=Cin, OH=OH, OW=OW, Cout=Cout, A=OH, B=OW,
                                          oy0=0, ox0=0, os=1, **{"is": stride}, ntaps=9, dy=dy, dx=dx, wi=wi, wtaps=9, x=x, w=wp, y=y,
                                          pro=cabi.pro(mode, scale, shift), residual=None, stats=st))
            dw = torch.zeros(Cout, Cin, 3, 3, device=be.device)
            be.call("conv_wgrad", cabi.make("mds_conv_wgrad_args", dtype=code, N=N, IH=H, IW=W, Cin=Cin, OH=OH, OW=OW, Cout=Cout,
                                            **{"is": stride}, ntaps=9, dy=dy, dx=dx, wi=wi, wtaps=9, x=x, dyt=dyt, dw=dw,
                                            pro=cabi.pro(mode, scale, shift)))
            be.sync()
            out[knob] = (y, st.sum(0), dw)
        finally:
            be.lib.fn["dev_set"](cabi.MDS_KNOB_C3, 0)
    (y0, s0, w0), (y1, s1, w1) = out[0], out[1]
    assert not torch.isnan(y0.float()).any()
    assert_close(y0, y1, "bf16", msg="y")
    cnt = N * OH * OW
    assert_close(s0[0], s1[0], "bf16", scale=cnt ** 0.5, msg="sum")
    assert_close(s0[1], s1[1], "bf16", scale=cnt ** 0.5, msg="sumsq")
    assert_close(w0, w1, "bf16", scale=cnt ** 0.5, msg="dw")


def _rand_cases(n, seed):
    import random
    r = random.Random(seed)
    out = []
    for _ in range(n):
        stride = r.choice([1, 1, 2])
        out.append((r.randint(1, 3), r.randint(5, 70), r.randint(5, 90), 8 * r.randint(1, 17), 16 * r.randint(1, 13), stride, r.choice([0, 1, 2]),
                    r.choice([0, 0, 3, 7])))
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("N,H,W,Cin,Cout,stride,mode,blocks", _rand_cases(16, 20260928))
def test_conv_fwd_random_shapes(be_gpu, dt, N, H, W, Cin, Cout, stride, mode, blocks):
    """seeded random layer shapes through whichever kernel the launcher picks (persistent: every N-tile / row-fragment
    template, K tails; chunked: Cin % 32 != 0, Cout tails), default grid and a few-block grid (GPU only: the simulator
    is too slow for 32 random layers)"""
    be = be_gpu
    be.lib.check(be.lib.fn["dev_set"](cabi.MDS_KNOB_CONV_BLOCKS, blocks), "dev_set")
    try:
        test_conv_fwd(be, dt, N, H, W, Cin, Cout, stride, mode)
    finally:
        be.lib.fn["dev_set"](cabi.MDS_KNOB_CONV_BLOCKS, 0)
