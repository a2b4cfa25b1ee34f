"""mds_pwg_fwd — the prologue-free, direct-to-LDS 1x1-convolution GEMM (bf16) — against a torch fp32 reference:
K / N / row tails, per-group weight sets, two operand pairs, bias, residual, forward statistics, post statistics."""
import pytest
import torch

from backends import be, assert_close  # noqa: F401
from mds import cabi

BF = torch.bfloat16


def pack(be, w, groups=1, kscale=None, nscale=None, transposed=False):
    """fp32 master [N][K] (or [K][N] when transposed) -> packed bf16 [groups][N][Kp] through mds_pwg_pack"""
    if transposed:
        K, N = w.shape
    else:
        N, K = w.shape
    Kp = (K + 63) // 64 * 64
    dst = torch.full((groups, N, Kp), float("nan"), dtype=BF, device=be.device)
    be.call("pwg_pack", cabi.make("mds_pwg_pack_args", src=be.t(w.float()), transposed=int(transposed), N=N, K=K, groups=groups,
                                  kscale=None if kscale is None else be.t(kscale), nscale=None if nscale is None else be.t(nscale), dst=dst))
    return dst


@pytest.mark.parametrize("M,groups,K0,N,K1,bias,res,stats", [
    (300, 1, 64, 128, 0, False, False, True),        # one full-width tile column, row tail
    (520, 2, 112, 112, 0, False, False, True),       # K tail (112 = 64 + 48), N below the tile, per-group weights, 2 tiles + tail per group
    (260, 1, 192, 400, 0, True, True, False),        # several n-tiles incl. a partial one (400 = 3 x 128 + 16 / 2 x 192 + 16), bias + residual
    (1000, 4, 48, 192, 96, True, False, True),       # two operand pairs, 192-wide tile, per-group weights on pair 0
    (77, 1, 1152, 16, 0, False, True, False),        # 18 k-chunks, single 16-column fragment
])
def test_pwg_fwd(be, M, groups, K0, N, K1, bias, res, stats):
    g = torch.Generator().manual_seed(M * 3 + N + K0)
    rpg = M // groups
    M = rpg * groups
    grp = torch.arange(M) // rpg
    x0 = torch.randn(M, K0, generator=g).to(BF)
    w0 = torch.randn(N, K0, generator=g) / K0 ** 0.5
    wg0 = groups > 1
    ks0 = torch.rand(groups, K0, generator=g) + 0.5 if wg0 else None
    w0p = pack(be, w0, groups if wg0 else 1, kscale=ks0)
    w0eff = (w0[None] * (ks0[:, None, :] if wg0 else 1.0)).to(BF).float()      # [groups or 1][N][K]
    ref = torch.einsum("mk,mnk->mn", x0.float(), w0eff[grp if wg0 else torch.zeros(M, dtype=torch.long)])
    kw = dict(x1=None, w1=None, K1=0, wg1=0)
    if K1:
        x1 = torch.randn(M, K1, generator=g).to(BF)
        w1 = torch.randn(K1, N, generator=g) / K1 ** 0.5            # stored transposed, like a data-gradient operand
        ns = torch.rand(N, generator=g) + 0.5
        w1p = pack(be, w1, 1, nscale=ns, transposed=True)
        ref = ref + x1.float() @ (w1.t() * ns[:, None]).to(BF).float().t()
        kw = dict(x1=be.t(x1), w1=w1p, K1=K1, wg1=0)
    b = torch.randn(N, generator=g) if bias else None
    r = torch.randn(M, N, generator=g).to(BF) if res else None
    if bias:
        ref = ref + b
    if res:
        ref = ref + r.float()
    y = torch.full((M, N), float("nan"), dtype=BF, device=be.device)
    st = torch.zeros(cabi.MDS_STAT_SLOTS, 2, N, device=be.device) if stats else None
    zeros = torch.zeros(64, device=be.device)
    be.call("pwg_fwd", cabi.make("mds_pwg_args", M=M, N=N, groups=groups, rows_per_group=rpg, npairs=2 if K1 else 1, x0=be.t(x0), K0=K0,
                                 w0=w0p, wg0=int(wg0), bias=None if b is None else be.t(b), y=y, residual=None if r is None else be.t(r),
                                 stats=st, zeros=zeros, **kw))
    be.sync()
    assert_close(y, ref, "bf16", msg="y")
    if stats:
        s = st.sum(0).cpu()
        assert_close(s[0], ref.sum(0), "bf16", scale=M ** 0.5, msg="sum")
        assert_close(s[1], (ref * ref).sum(0), "bf16", scale=M ** 0.5, msg="sumsq")


@pytest.mark.parametrize("mode", [1, 2, 3])
def test_pwg_fwd_post_statistics(be, mode):
    """data-gradient use: BatchNorm-backward sums of the next layer over the output tile (PLAIN / MASK / SILU)"""
    g = torch.Generator().manual_seed(40 + mode)
    M, K, N, groups = 600, 96, 48, 3
    rpg = M // groups
    grp = torch.arange(M) // rpg
    x = torch.randn(M, K, generator=g).to(BF)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    r = torch.randn(M, N, generator=g).to(BF)
    ys = (1.2 * torch.randn(M, N, generator=g) - 0.2).to(BF)
    bn = torch.stack([1 + 0.2 * torch.randn(N, generator=g), 0.1 * torch.randn(N, generator=g), 0.3 * torch.randn(N, generator=g),
                      0.5 + torch.rand(N, generator=g)])
    mask = (torch.rand(groups, generator=g) < 0.6).float() / 0.6
    y = torch.full((M, N), float("nan"), dtype=BF, device=be.device)
    st = torch.zeros(cabi.MDS_STAT_SLOTS, 2, N, device=be.device)
    zeros = torch.zeros(64, device=be.device)
    post = cabi.poststat(mode, be.t(ys), be.t(bn), st, be.t(mask), rpg)
    be.call("pwg_fwd", cabi.make("mds_pwg_args", M=M, N=N, groups=1, rows_per_group=M, npairs=1, x0=be.t(x), K0=K, w0=pack(be, w), wg0=0,
                                 x1=None, w1=None, K1=0, wg1=0, bias=None, y=y, residual=be.t(r), stats=None, post=post, zeros=zeros))
    be.sync()
    v = x.float() @ w.to(BF).float().t() + r.float()
    zs = ys.float() * bn[0] + bn[1]
    sg = torch.sigmoid(zs)
    stored = v * (sg * (1 + zs * (1 - sg))) if mode == 3 else v
    assert_close(y, stored, "bf16", scale=2, msg="stored")
    gq = y.float().cpu() * (mask[grp, None] if mode == 2 else 1.0)
    xh = (ys.float() - bn[2]) * bn[3]
    s = st.sum(0).cpu()
    assert_close(s[0], gq.sum(0), "f32", scale=50 * M ** 0.5, msg="sum g")
    assert_close(s[1], (gq * xh).sum(0), "f32", scale=50 * M ** 0.5, msg="sum g*xhat")
