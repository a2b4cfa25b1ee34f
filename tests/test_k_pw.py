"""1x1-conv GEMM kernels (mds_pw_fwd / mds_pw_wgrad) against a torch fp32 reference."""
import pytest
import torch
import torch.nn.functional as F

from backends import be, be_gpu, DT, assert_close  # noqa: F401
from mds import cabi


def _mk_pro(be, mode, K, groups, g):
    scale = be.t(1.0 + 0.2 * torch.randn(K, generator=g))
    shift = be.t(0.3 * torch.randn(K, generator=g))
    gate = be.t(torch.rand(groups, K, generator=g))
    return scale, shift, gate


def _apply_pro(x, mode, scale, shift, gate, rpg):
    if mode == 0:
        return x
    if mode == 4:                      # MDS_PRO_GATE: x is the materialised activation, gate only
        return x * gate.cpu()[torch.arange(x.shape[0]) // rpg]
    z = x * scale.cpu() + shift.cpu()
    if mode == 1:
        return z
    a = F.silu(z)
    if mode == 3:
        M = x.shape[0]
        grp = torch.arange(M) // rpg
        a = a * gate.cpu()[grp]
    return a


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("M,K,N,mode,res,stats", [
    (300, 32, 64, 0, False, True),
    (260, 48, 144, 2, False, True),     # K tail (48 = 32 + 16), two n-tiles (128 + 16)
    (200, 112, 32, 3, False, True),
    (130, 16, 16, 1, True, False),
    (257, 192, 192, 0, True, False),
    (150, 576, 48, 3, True, True),      # K >= 512 (bf16) / 256 (fp32): 512-byte K chunks, with a K tail (576 = 4.5 x 128)
    (150, 1160, 192, 0, False, True),   # ... two n-tiles, K % 64 != 0
    (300, 320, 128, 4, False, True),    # gated projection, 5 K chunks: the two-chunk ring with its phantom chunk
    (200, 704, 192, 4, True, False),    # ... 11 chunks, two n-tiles, residual
    (400300, 16, 144, 2, False, True),  # above 400 k rows: the 128-row tile path (GPU only: too slow to simulate)
])
def test_pw_fwd(be, dt, M, K, N, mode, res, stats):
    if M > 100000 and be.name == "emu":
        pytest.skip("large-M tile path is exercised on the GPU")
    code, tdt = DT[dt]
    g = torch.Generator().manual_seed(M * 7 + K)
    rpg = 50
    groups = (M + rpg - 1) // rpg
    x = torch.randn(M, K, generator=g).to(tdt)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(tdt)
    r = torch.randn(M, N, generator=g).to(tdt)
    scale, shift, gate = _mk_pro(be, mode, K, groups, g)
    xd, wd, rd = be.t(x), be.t(w), be.t(r)
    y = torch.full((M, N), float("nan")).to(tdt).to(be.device)
    st = torch.zeros(cabi.MDS_STAT_SLOTS, 2, N, device=be.device, dtype=torch.float64)
    args = cabi.make("mds_pw_fwd_args", dtype=code, M=M, K=K, N=N, x=xd, w=wd, y=y,
                     pro=cabi.pro(mode, scale, shift, gate, rpg),
                     residual=rd if res else None, stats=st if stats else None)
    be.call("pw_fwd", args)
    be.sync()
    a = _apply_pro(x.float(), mode, scale, shift, gate, rpg)
    if dt == "bf16":
        a = a.to(tdt).float()
    ref = a @ w.float().t()
    if res:
        ref = ref + r.float()
    assert_close(y, ref, dt, msg="y")
    if stats:
        s = st.sum(0).cpu()
        assert_close(s[0], ref.sum(0), dt, scale=M ** 0.5, msg="sum")
        assert_close(s[1], (ref * ref).sum(0), dt, scale=M ** 0.5, msg="sumsq")


@pytest.mark.parametrize("M,K,N,res,blocks", [(32 * 20, 32, 128, False, 0), (640 * 7, 32, 128, True, 2), (96 * 11, 32, 64, False, 0), (64 * 9, 48, 192, False, 1),
                                             (160 * 6, 48, 128, True, 3)])
def test_pw_fwd_through_the_row_streaming_kernel(be, M, K, N, res, blocks):
    """the large prologue-free bf16 1x1 launches (the edge-residual projections' data gradients) go through k_c3.hip as one-tap
    'images' of W-pixel rows (c3_pw_try); MDS_KNOB_C3 = 2 lifts the 262 144-row bar"""
    be.lib.check(be.lib.fn["dev_set"](cabi.MDS_KNOB_C3, 2), "dev_set")
    be.lib.check(be.lib.fn["dev_set"](cabi.MDS_KNOB_CONV_BLOCKS, blocks), "dev_set")
    try:
        test_pw_fwd(be, "bf16", M, K, N, 0, res, False)
    finally:
        be.lib.fn["dev_set"](cabi.MDS_KNOB_C3, 0)
        be.lib.fn["dev_set"](cabi.MDS_KNOB_CONV_BLOCKS, 0)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("M,K,N,pmode,emode,res", [(300, 48, 144, 0, 2, False), (200, 112, 32, 4, 1, True), (130, 192, 192, 0, 2, False),
                                                   (150, 704, 192, 4, 1, True), (140, 320, 128, 0, 2, False)])   # K-heavy inference shapes (the predictor's one-image plans)
def test_pw_fwd_output_transform(be, dt, M, K, N, pmode, emode, res):
    """mds_epi_t: y = act(acc*scale + shift) (+ residual) - the inference plans' producers store activated outputs"""
    code, tdt = DT[dt]
    g = torch.Generator().manual_seed(M + N + emode)
    rpg = 97
    groups = (M + rpg - 1) // rpg
    x = torch.randn(M, K, generator=g).to(tdt)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(tdt)
    scale, shift, gate = _mk_pro(be, pmode, K, groups, g)
    esc = 1.0 + 0.3 * torch.randn(N, generator=g); esh = 0.5 * torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g).to(tdt) if res else None
    y = torch.full((M, N), float("nan")).to(tdt).to(be.device)
    be.call("pw_fwd", cabi.make("mds_pw_fwd_args", dtype=code, M=M, K=K, N=N, x=be.t(x), w=be.t(w), y=y,
                                pro=cabi.pro(pmode, scale, shift, gate, rpg), residual=be.t(r) if res else None, stats=None,
                                epi=cabi.make("mds_epi_t", mode=emode, scale=be.t(esc), shift=be.t(esh))))
    be.sync()
    a = _apply_pro(x.float(), pmode, scale, shift, gate, rpg)
    if dt == "bf16":
        a = a.to(tdt).float()
    ref = (a @ w.float().t()) * esc + esh
    if emode == 2:
        ref = F.silu(ref)
    if res:
        ref = ref + r.float()
    assert_close(y, ref, dt, msg="y")


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("M,K,N,pmode,emode,res,split", [(300, 448, 144, 0, 2, False, 3), (200, 672, 112, 4, 1, True, 5), (130, 1152, 192, 4, 1, True, 16),
                                                      (77, 200, 48, 0, 0, False, 2), (920, 360, 192, 4, 1, False, 4)])
def test_pw_fwd_split_k(be, dt, M, K, N, pmode, emode, res, split):
    """split-K of the small-M inference GEMMs (mds_pw_fwd_args.split): partial tiles + ticket, the last block of a tile adds the
    partials in z order and runs the epilogue.  Same result as the unsplit launch to rounding, bit-identical between two split
    launches (the order of the sum is fixed), tickets back at zero afterwards, K ranges with a ragged / empty last split."""
    code, tdt = DT[dt]
    g = torch.Generator().manual_seed(M + N + split)
    rpg = 97
    groups = (M + rpg - 1) // rpg
    x = torch.randn(M, K, generator=g).to(tdt)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(tdt)
    scale, shift, gate = _mk_pro(be, pmode, K, groups, g)
    esc = 1.0 + 0.3 * torch.randn(N, generator=g); esh = 0.5 * torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g).to(tdt) if res else None
    part = torch.full((split * M * N,), float("nan"), device=be.device)
    tiles = -(-M // cabi.MDS_PW_SPLIT_TILE_ROWS) * -(-N // 128)
    ticket = torch.zeros(tiles * cabi.MDS_PW_SPLIT_TICKET_STRIDE, dtype=torch.int32, device=be.device)
    outs = []
    for sp in (split, split, 0):
        y = torch.full((M, N), float("nan")).to(tdt).to(be.device)
        epi = cabi.make("mds_epi_t", mode=emode, scale=be.t(esc), shift=be.t(esh)) if emode else cabi.make("mds_epi_t", mode=0)
        be.call("pw_fwd", cabi.make("mds_pw_fwd_args", dtype=code, M=M, K=K, N=N, x=be.t(x), w=be.t(w), y=y,
                                    pro=cabi.pro(pmode, scale, shift, gate, rpg), residual=be.t(r) if res else None, stats=None, epi=epi,
                                    split=sp, split_part=part if sp else None, split_ticket=ticket if sp else None))
        be.sync()
        outs.append(y.float().cpu())
        assert int(ticket.abs().sum()) == 0, "tickets must reset themselves"
    a = _apply_pro(x.float(), pmode, scale, shift, gate, rpg)
    if dt == "bf16":
        a = a.to(tdt).float()
    ref = a @ w.float().t()
    if emode:
        ref = ref * esc + esh
    if emode == 2:
        ref = F.silu(ref)
    if res:
        ref = ref + r.float()
    assert_close(outs[0], ref, dt, msg="y (split)")
    assert torch.equal(outs[0], outs[1]), "two split launches must agree bit for bit"
    assert_close(outs[0], outs[2], dt, msg="split vs unsplit")


@pytest.mark.gpu
def test_pw_fwd_split_k_handoff_under_load(be_gpu):
    """the in-launch hand-off of the partial tiles (device-scope stores, barrier, ticket, device-scope loads - no fence) on the
    hardware it was designed on: every launch reproduces the first bit for bit while a second stream streams copies through
    HBM and the partial buffer is NaN-poisoned between launches (a stale or early read would be a NaN)"""
    b = be_gpu
    dev = b.device
    side = torch.cuda.Stream()
    big = torch.empty(64 << 20, dtype=torch.float32, device=dev); big2 = torch.empty_like(big)
    for (M, K, N, S, code, tdt) in [(920, 1152, 192, 12, cabi.MDS_F32, torch.float32), (3680, 672, 112, 4, cabi.MDS_BF16, torch.bfloat16)]:
        x = torch.randn(M, K, device=dev).to(tdt); w = (torch.randn(N, K, device=dev) / K ** 0.5).to(tdt)
        y = torch.empty(M, N, device=dev, dtype=tdt)
        part = torch.empty(S * M * N, device=dev)
        tk = torch.zeros(-(-M // cabi.MDS_PW_SPLIT_TILE_ROWS) * -(-N // 128) * cabi.MDS_PW_SPLIT_TICKET_STRIDE, dtype=torch.int32, device=dev)
        a = cabi.make("mds_pw_fwd_args", dtype=code, M=M, K=K, N=N, x=x, w=w, y=y, pro=cabi.pro(0), residual=None, stats=None,
                      split=S, split_part=part, split_ticket=tk)
        b.call("pw_fwd", a); b.sync()
        ref = y.clone()
        assert torch.isfinite(ref.float()).all()
        for it in range(300):
            if it % 25 == 0:
                with torch.cuda.stream(side):
                    big2.copy_(big)
            part.fill_(float("nan"))
            b.call("pw_fwd", a)
            assert torch.equal(y, ref), it
        assert int(tk.abs().sum()) == 0


def test_pw_fwd_split_rule(be):
    f = be.lib.fn["pw_fwd_split"]
    assert f(920, 1152, 192, cabi.MDS_F32) >= 2 and f(3680, 672, 112, cabi.MDS_F32) >= 2       # the gated projections of one 736x1280 frame
    assert f(18400, 1152, 192, cabi.MDS_BF16) == 1 and f(920, 192, 1152, cabi.MDS_F32) == 1     # training shapes / short K: never
    assert all(1 <= f(m, k, n, d) <= cabi.MDS_PW_MAX_SPLIT for m in (1, 64, 920, 7360) for k in (8, 192, 1152) for n in (16, 192, 1280) for d in (0, 1))


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("M,K,N,mode", [
    (500, 32, 64, 0),
    (333, 48, 144, 2),
    (700, 112, 32, 3),
    (64, 192, 16, 1),
    (300, 200, 192, 4),     # 128 < N <= 256: a full and a half-empty 128-column tile, gate prologue, K tail
    (900, 136, 192, 3),     # the stage-6 projection's form: BN + SiLU + gate on the wide operand, K tail
    (450, 64, 160, 0),      # ragged second tile
])
def test_pw_wgrad(be, dt, M, K, N, mode):
    code, tdt = DT[dt]
    g = torch.Generator().manual_seed(M * 3 + N)
    rpg = 97
    groups = (M + rpg - 1) // rpg
    x = torch.randn(M, K, generator=g).to(tdt)
    dy = torch.randn(M, N, generator=g).to(tdt)
    scale, shift, gate = _mk_pro(be, mode, K, groups, g)
    dw = torch.zeros(N, K, device=be.device)
    args = cabi.make("mds_pw_wgrad_args", dtype=code, M=M, K=K, N=N, x=be.t(x), dy=be.t(dy), dw=dw,
                     pro=cabi.pro(mode, scale, shift, gate, rpg))
    be.call("pw_wgrad", args)
    be.sync()
    a = _apply_pro(x.float(), mode, scale, shift, gate, rpg)
    if dt == "bf16":
        a = a.to(tdt).float()
    ref = dy.float().t() @ a
    assert_close(dw, ref, dt, scale=M ** 0.5, msg="dw")


def _bn_setup(be, y, gamma, beta, eps=1e-5):
    """forward statistics of a train-mode BN over raw input y -> the kernels' [4][C] buffer (scale, shift, mean, rstd)"""
    M, C = y.shape
    st = torch.zeros(cabi.MDS_STAT_SLOTS, 2, C, dtype=torch.float64)
    st[0, 0] = y.float().sum(0); st[0, 1] = (y.float() ** 2).sum(0)
    out = torch.empty(4, C, device=be.device)
    be.call("bn_finalize", cabi.make("mds_bn_finalize_args", C=C, count=M, stats=be.t(st), gamma=be.t(gamma), beta=be.t(beta), eps=eps,
                                     momentum=0.1, training=1, running_mean=None, running_var=None, num_batches_tracked=None, out=out))
    return out


def _bn_bwd_lin(be, g, y, bn, gamma):
    """sum g, sum g*xhat (torch) -> mds_bn_bwd_finalize -> lin = {A, B, D}"""
    M, C = y.shape
    bnc = bn.cpu()
    xhat = (y.float() - bnc[2]) * bnc[3]
    st = torch.zeros(cabi.MDS_STAT_SLOTS, 2, C, dtype=torch.float64)          # backward sums: fp64 slots
    st[3, 0] = g.sum(0); st[5, 1] = (g * xhat).sum(0)
    coef = torch.empty(3, C, device=be.device); lin = torch.empty(3, C, device=be.device)
    be.call("bn_bwd_finalize", cabi.make("mds_bn_bwd_finalize_args", C=C, count=M, stats=be.t(st), gamma=be.t(gamma), bn=bn, dgamma=None,
                                         dbeta=None, coef=coef, lin=lin, batch_stats=1))
    return lin


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("M,C,N,gmode,res,post", [
    (300, 48, 64, 0, False, 0),       # no post statistics
    (260, 96, 144, 3, True, 2),       # MASK gradient source + residual + MASK post statistics, two n-tiles
    (515, 16, 32, 0, True, 3),        # narrow (128x64 tile variant) + SILU post statistics (g stored)
    (130, 192, 16, 3, False, 1),      # several k-chunks, PLAIN post statistics
])
def test_pw_fwd_bn_backward_fusion(be, dt, M, C, N, gmode, res, post):
    """data-gradient GEMM whose epilogue takes the sums of the NEXT BatchNorm backward over the output tile (mds_poststat_t); its
    operand is the materialised dy = BN'(g) of this layer (mds_bn_bwd_apply's result)."""
    code, tdt = DT[dt]
    g_ = torch.Generator().manual_seed(M + 13 * C + post)
    rpg = 37
    groups = (M + rpg - 1) // rpg
    grp = torch.arange(M) // rpg
    u = torch.randn(M, C, generator=g_).to(tdt)
    y = (1.5 * torch.randn(M, C, generator=g_) + 0.3).to(tdt)
    gamma = 1 + 0.2 * torch.randn(C, generator=g_); beta = 0.1 * torch.randn(C, generator=g_)
    mask = (torch.rand(groups, generator=g_) < 0.7).float() / 0.7
    w = (torch.randn(N, C, generator=g_) / C ** 0.5).to(tdt)
    r = torch.randn(M, N, generator=g_).to(tdt)
    # reference BN backward (fp32 autograd on the stored values)
    yf = y.float().requires_grad_(True)
    z = F.batch_norm(yf, None, None, gamma, beta, True, 0.1, 1e-5)
    gsrc_ref = u.float() * (mask[grp, None] if gmode == 3 else 1.0)
    z.backward(gsrc_ref)
    dy_ref = yf.grad
    dyd = be.t(dy_ref.to(tdt))
    out = torch.full((M, N), float("nan")).to(tdt).to(be.device)
    # next layer's BatchNorm (the one whose backward sums the epilogue takes)
    ys = (torch.randn(M, N, generator=g_) * 1.2 - 0.2).to(tdt)
    gamma2 = 1 + 0.2 * torch.randn(N, generator=g_); beta2 = 0.1 * torch.randn(N, generator=g_)
    mask2 = (torch.rand(groups, generator=g_) < 0.6).float() / 0.6
    bn2 = _bn_setup(be, ys, gamma2, beta2)
    st2 = torch.zeros(cabi.MDS_STAT_SLOTS, 2, N, device=be.device, dtype=torch.float64)
    kw = {}
    if post:
        kw["post"] = cabi.poststat(post, be.t(ys), bn2, st2, be.t(mask2), rpg)
    be.call("pw_fwd", cabi.make("mds_pw_fwd_args", dtype=code, M=M, K=C, N=N, x=dyd, w=be.t(w), y=out, pro=cabi.pro(0),
                                residual=be.t(r) if res else None, stats=None, **kw))
    be.sync()
    dyq = dy_ref.to(tdt).float() if dt == "bf16" else dy_ref
    v = dyq @ w.float().t() + (r.float() if res else 0.0)
    b2 = bn2.cpu()
    zs = ys.float() * b2[0] + b2[1]
    sg = torch.sigmoid(zs)
    stored = v * (sg * (1 + zs * (1 - sg))) if post == 3 else v
    assert_close(out, stored, dt, scale=2, msg="out")
    if post:
        gq = out.float().cpu() * (mask2[grp, None] if post == 2 else 1.0)      # the sums are defined on what was stored
        xh = (ys.float() - b2[2]) * b2[3]
        s = st2.sum(0).cpu()
        assert_close(s[0], gq.sum(0), "f32", scale=50 * M ** 0.5, msg="post sum g")
        assert_close(s[1], (gq * xh).sum(0), "f32", scale=50 * M ** 0.5, msg="post sum g*xhat")


@pytest.fixture
def force_filter_resident(be):
    be.lib.check(be.lib.fn["dev_set"](cabi.MDS_KNOB_PW_WRES, 2), "dev_set")
    yield
    be.lib.fn["dev_set"](cabi.MDS_KNOB_PW_WRES, 0)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("M,K,N,res,stats,post", [
    (1300, 192, 256, False, True, 0),    # 6 k-steps, 3 row tiles per block, ragged last tile
    (650, 112, 192, False, True, 0),     # K % 32 != 0 (zero k-padding), 96-column tiles (N % 96 == 0)
    (530, 96, 144, False, True, 0),      # ragged n-tile (128 + 16)
    (600, 96, 128, False, False, 1),     # PLAIN post statistics
    (1290, 192, 192, False, False, 2),   # MASK post statistics
    (515, 80, 256, False, False, 3),     # SILU post statistics (g stored), K = 80 padded to 96
    (2500, 96, 128, False, True, 0),     # 5 row tiles per block: the 3-deep register ring wraps and refills
    (2300, 112, 192, False, False, 2),   # ... with the post.y fragments prefetched one tile ahead
    (2560, 96, 144, False, True, 0),     # no ragged row tile: only the straight-line trips, plus a ragged n-tile block
])
def test_pw_fwd_filter_resident(be, force_filter_resident, dt, M, K, N, res, stats, post):
    """k_pwr.hip: the short-K / wide-N kernel (filter tile resident in LDS, column sums once per block).  Launches with post
    statistics take the GENERAL kernel since round 4 (that form spilled 137-667 VGPRs and no layer reaches it) - same checks."""
    if dt == "f32" and K > 96:
        pytest.skip("fp32 filter tile of K > 96 does not fit two blocks per CU: general kernel")
    _run_pw_plain(be, dt, M, K, N, res, stats, post, post == 0)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("M,K,N,res,stats,post", [
    (300, 320, 128, False, True, 0),     # 5 chunks (bf16): odd count -> one all-zero phantom chunk
    (200, 704, 192, True, False, 1),     # 11 chunks, two n-tiles, residual, PLAIN post statistics
    (130, 1096, 144, False, False, 3),   # K % 64 != 0, ragged n-tile, SILU post statistics
])
def _run_pw_plain(be, dt, M, K, N, res, stats, post, check_taken, frag=False, form=0):
    code, tdt = DT[dt]
    g_ = torch.Generator().manual_seed(M + 3 * K + post)
    rpg = 41
    groups = (M + rpg - 1) // rpg
    grp = torch.arange(M) // rpg
    x = torch.randn(M, K, generator=g_).to(tdt)
    w = (torch.randn(N, K, generator=g_) / K ** 0.5).to(tdt)
    r = torch.randn(M, N, generator=g_).to(tdt)
    out = torch.full((M, N), float("nan")).to(tdt).to(be.device)
    st = torch.zeros(cabi.MDS_STAT_SLOTS, 2, N, device=be.device, dtype=torch.float64)
    kw = {}
    if post:
        ys = (torch.randn(M, N, generator=g_) * 1.2 - 0.2).to(tdt)
        gamma2 = 1 + 0.2 * torch.randn(N, generator=g_); beta2 = 0.1 * torch.randn(N, generator=g_)
        mask2 = (torch.rand(groups, generator=g_) < 0.6).float() / 0.6
        bn2 = _bn_setup(be, ys, gamma2, beta2)
        kw["post"] = cabi.poststat(post, be.t(ys), bn2, st, be.t(mask2), rpg)
    if frag:
        kw["w_frag"] = _frag_pack(be, w, io=True)
    be.call("pw_fwd", cabi.make("mds_pw_fwd_args", dtype=code, M=M, K=K, N=N, x=be.t(x), w=be.t(w), y=out, pro=cabi.pro(0),
                                residual=be.t(r) if res else None, stats=st if stats else None, form=form, **kw))
    be.sync()
    if frag and post and form != 1:
        assert int((st.abs().sum((1, 2)) > 0).sum()) <= -(-M // 64), "the K-streaming kernel was not taken"
    v = x.float() @ w.float().t() + (r.float() if res else 0.0)
    if check_taken and (stats or post) and M > 640:   # this kernel adds into 8 + 1 statistic slots (8 blocks per n-tile), the general one into M/64 + 1
        assert int((st.abs().sum((1, 2)) > 0).sum()) <= 9, "the filter-resident kernel was not taken"
    if post:
        b2 = bn2.cpu()
        zs = ys.float() * b2[0] + b2[1]
        sg = torch.sigmoid(zs)
        stored = v * (sg * (1 + zs * (1 - sg))) if post == 3 else v
        assert_close(out, stored, dt, scale=2, msg="out")
        gq = out.float().cpu() * (mask2[grp, None] if post == 2 else 1.0)
        xh = (ys.float() - b2[2]) * b2[3]
        s = st.sum(0).cpu()
        assert_close(s[0], gq.sum(0), "f32", scale=50 * M ** 0.5, msg="post sum g")
        assert_close(s[1], (gq * xh).sum(0), "f32", scale=50 * M ** 0.5, msg="post sum g*xhat")
    else:
        assert_close(out, v, dt, msg="y")
        if stats:
            s = st.sum(0).cpu()
            assert_close(s[0], v.sum(0), dt, scale=M ** 0.5, msg="sum")
            assert_close(s[1], (v * v).sum(0), dt, scale=M ** 0.5, msg="sumsq")


def _frag_pack(be, w, io=False):
    """the fragment-major bf16 copy of a 1x1 filter through mds_pack_weights (w: [N][K]; io: packed from its transpose, as the
    data-gradient launches do from the [K][N] parameter)"""
    N, K = w.shape
    src = be.t(w.float().t() if io else w.float())
    dst = torch.empty(-(-K // 32) * -(-N // 16) * 512, dtype=torch.bfloat16, device=be.device)
    Job = cabi.STRUCTS["mds_pack_job"]
    job = Job()
    job.src, job.dst = src.data_ptr(), dst.data_ptr()
    job.kind = cabi.MDS_PACK_FRAG_IO if io else cabi.MDS_PACK_FRAG_OI
    job.O, job.I, job.taps = (K, N, 1) if io else (N, K, 1)
    tab = torch.frombuffer(bytearray(bytes(job)), dtype=torch.uint8).to(be.device)
    be.lib.check(be.lib.fn["pack_weights"](tab.data_ptr(), 1, dst.numel(), cabi.MDS_BF16, be.stream()), "pack_weights")
    be.sync()
    return dst


def test_frag_pack_layout(be):
    """MDS_PACK_FRAG_OI / _IO against the index formula of include/mds.h, with N and K padding"""
    g = torch.Generator().manual_seed(5)
    N, K = 40, 72
    w = torch.randn(N, K, generator=g).to(torch.bfloat16)
    NFT, KST = -(-N // 16), -(-K // 32)
    want = torch.zeros(KST * NFT * 512)
    n, k = torch.meshgrid(torch.arange(N), torch.arange(K), indexing="ij")
    idx = ((k // 32 * NFT + n // 16) * 64 + (n % 16) + 16 * (k % 32 // 8)) * 8 + k % 8
    want[idx.flatten()] = w.float().flatten()
    for io in (False, True):
        assert torch.equal(_frag_pack(be, w, io).float().cpu(), want), io


@pytest.fixture
def force_kstream(be):
    be.lib.check(be.lib.fn["dev_set"](cabi.MDS_KNOB_PWK, 2), "dev_set")
    yield
    be.lib.fn["dev_set"](cabi.MDS_KNOB_PWK, 0)


@pytest.mark.parametrize("M,K,N,mode,stats", [
    (300, 64, 192, 0, True),       # one stage only: the rings are longer than the stream
    (333, 1152, 192, 3, True),     # the stage-5 projection: BN + SiLU + gate, 18 stages, ragged last row tile, a gate boundary inside a tile
    (200, 672, 112, 3, True),      # 128-column tile with 16 padding columns; K % 64 == 32: a half stage at the end
    (260, 384, 96, 3, True),       # 96-column tile of 128 rows (two row blocks per wave column, two transform passes)
    (150, 576, 192, 4, False),     # gate only, no statistics
    (140, 192, 176, 2, True),      # BN + SiLU, N not a multiple of 48
    (130, 160, 80, 1, True),       # affine prologue, 96-column tile with a padding fragment, half stage
    (1000, 256, 128, 0, True),
    (70, 96, 128, 2, True),        # a half stage right after the first
])
def test_pw_fwd_kstream(be, force_kstream, M, K, N, mode, stats):
    """k_pwk8.hip: the K-streaming kernel (x as an LDS-DMA ring with counted waits, prologue two stages ahead through scalar
    tables, row-major epilogue).  On the simulator this checks addressing and slot arithmetic; the waits on MI355X."""
    _kstream_case(be, M, K, N, mode, stats)


def _kstream_case(be, M, K, N, mode, stats):
    dt = "bf16"
    code, tdt = DT[dt]
    g = torch.Generator().manual_seed(M * 7 + K + N)
    rpg = 150
    groups = (M + rpg - 1) // rpg
    x = torch.randn(M, K, generator=g).to(tdt)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(tdt)
    scale, shift, gate = _mk_pro(be, mode, K, groups, g)
    y = torch.full((M, N), float("nan")).to(tdt).to(be.device)
    st = torch.zeros(cabi.MDS_STAT_SLOTS, 2, N, device=be.device, dtype=torch.float64)
    be.call("pw_fwd", cabi.make("mds_pw_fwd_args", dtype=code, M=M, K=K, N=N, x=be.t(x), w=be.t(w), y=y,
                                pro=cabi.pro(mode, scale, shift, gate, rpg), residual=None, stats=st if stats else None,
                                w_frag=_frag_pack(be, w, io=(M % 2 == 1))))
    be.sync()
    a = _apply_pro(x.float(), mode, scale, shift, gate, rpg).to(tdt).float()
    ref = a @ w.float().t()
    assert_close(y, ref, dt, msg="y")
    if stats:
        # this kernel adds one partial per 64- / 128-row block: at most ceil(M / 64) slots are touched
        assert int((st.abs().sum((1, 2)) > 0).sum()) <= -(-M // 64), "the K-streaming kernel was not taken"
        s = st.sum(0).cpu()
        assert_close(s[0], ref.sum(0), dt, scale=M ** 0.5, msg="sum")
        assert_close(s[1], (ref * ref).sum(0), dt, scale=M ** 0.5, msg="sumsq")


@pytest.mark.parametrize("bm", [80, 96, 128])
@pytest.mark.parametrize("M,K,N,mode", [
    (333, 1152, 192, 3),           # rows_per_group 150: gate boundaries in the first and in the second 64-row transform pass
    (290, 672, 112, 3),            # 128-column tile, half stage
    (200, 576, 192, 0),            # no prologue: the producers only stream
    (170, 96, 128, 2),
])
def test_pw_fwd_kstream_tile_rows(be, force_kstream, bm, M, K, N, mode):
    """k_pwk8.hip: the other row counts of a tile (80 / 96 / 128: three or four DMA blocks per producer wave, a partial second
    transform pass, 5 / 6 / 8 row fragments per consumer)"""
    be.lib.check(be.lib.fn["dev_set"](cabi.MDS_KNOB_PWK_BM, bm), "dev_set")
    try:
        _kstream_case(be, M, K, N, mode, True)
    finally:
        be.lib.fn["dev_set"](cabi.MDS_KNOB_PWK_BM, 0)


@pytest.mark.parametrize("M,K,N,res,post", [
    (300, 1152, 192, True, 2),     # the stage-5 expansion's data gradient: residual + MASK post statistics
    (210, 672, 112, True, 1),      # PLAIN
    (260, 384, 96, False, 3),      # SILU (g stored), 128-row tiles
    (150, 576, 192, True, 0),      # residual only
    (140, 192, 144, False, 1),     # N = 144: 18 octets per row
    (300, 96, 80, True, 2),        # one and a half stages
])
def test_pw_fwd_kstream_data_gradient(be, force_kstream, M, K, N, res, post):
    _run_pw_plain(be, "bf16", M, K, N, res, False, post, False, frag=True)


@pytest.mark.parametrize("form,res,post", [(2, False, 0), (2, True, 2), (1, True, 0), (1, False, 0)])
def test_pw_fwd_form_field_decides_what_the_kstream_kernel_sees(be, force_kstream, form, res, post):
    """mds_pw_fwd_args.form: the planner's own forward / data-gradient flag instead of inferring it from the operands - a
    data gradient without residual and post sums still takes the data-gradient tiles, a FORWARD launch with a fused residual is not
    mistaken for one (it goes to the general kernel); every combination gives the same numbers"""
    _run_pw_plain(be, "bf16", 300, 1152, 192, res, False, post, False, frag=True, form=form)


# ------------------------------------------------------------------------------------------------ linear form of BatchNorm backward
def _wcat(w0, w1):
    """[N][K0] and [N][K1] -> the packed [N][Kp + K1p] weight rows of a two-pair mds_pw_fwd (zero padded to multiples of 64)"""
    N, K0 = w0.shape
    K1 = w1.shape[1]
    Kp, K1p = (K0 + 63) // 64 * 64, (K1 + 63) // 64 * 64
    out = torch.zeros(N, Kp + K1p, dtype=w0.dtype)
    out[:, :K0] = w0
    out[:, Kp:Kp + K1] = w1
    return out

