"""mds_pw_dgrad (k_pwd.hip): the data gradient of a 1x1 expansion convolution with the BatchNorm-backward apply pass folded in,
against a float64 restatement of torch's native_batch_norm_backward (as dy = A*g + B*y + D) + the input half of
convolution_backward (/root/reference/src/models/multidim_stacker.py:124-134), on the simulator and on MI355X."""
import pytest
import torch

from backends import be, be_gpu, DT, assert_close  # noqa: F401
from mds import cabi

SLOTS = cabi.MDS_STAT_SLOTS


def gen(s):
    return torch.Generator().manual_seed(s)


def _run(be, dt, M, K, N, dyp, store, res, post, seed=0):
    code, tdt = DT[dt]
    g_ = gen(seed + M + K + N)
    u = torch.randn(M, K, generator=g_).to(tdt)
    y = (torch.randn(M, K, generator=g_) * 1.5 + 0.3).to(tdt)
    w = (torch.randn(N, K, generator=g_) / K ** 0.5).to(tdt)
    lin = torch.stack([0.5 + torch.rand(K, generator=g_), 0.2 * torch.randn(K, generator=g_), 0.1 * torch.randn(K, generator=g_)]).contiguous()
    resid = torch.randn(M, N, generator=g_).to(tdt) if res else None
    rpg = 37
    py = torch.randn(M, N, generator=g_).to(tdt)
    pbn = torch.stack([torch.ones(N), torch.zeros(N), 0.1 * torch.randn(N, generator=g_), 0.5 + torch.rand(N, generator=g_)]).contiguous()
    mask = (torch.rand((M + rpg - 1) // rpg, generator=g_) > 0.3).float() / 0.7
    stats = torch.zeros(SLOTS, 2, N, dtype=torch.float64, device=be.device)
    out = torch.full((M, N), float("nan")).to(tdt).to(be.device)
    dyo = torch.full((M, K), float("nan")).to(tdt).to(be.device) if (dyp and store) else None
    kw = dict(dtype=code, M=M, K=K, N=N, w=be.t(w), y=out, residual=be.t(resid) if res else None, dy_out=dyo)
    if dyp:
        kw["dyp"] = cabi.dyp(cabi.gsrc(0, be.t(u)), be.t(y), be.t(torch.zeros(4, K)), be.t(lin))
    else:
        kw["x"] = be.t(u)
    if post:
        kw["post"] = cabi.poststat(post, be.t(py), be.t(pbn), stats, mask=be.t(mask) if post == 2 else None, rows_per_group=rpg if post == 2 else 0)
    assert be.lib.fn["pw_dgrad_ok"](M, K, N) == 1
    be.call("pw_dgrad", cabi.make("mds_pw_dgrad_args", **kw))
    be.sync()
    A, B, D = lin.double()
    dy = (A * u.double() + B * y.double() + D) if dyp else u.double()
    if dyo is not None:
        assert_close(dyo, dy, dt, msg="stored dy")
        dy = dyo.cpu().double()            # what the kernel multiplies is what it stored (same rounding)
    elif dyp:
        dy = dy.to(tdt).double()
    ref = dy @ w.double().t()
    if res:
        ref = ref + resid.double()
    if dyp and dyo is None and dt == "bf16":
        # (no stored dy to take the rounding from: one bf16 ulp on the operands that sit at a rounding boundary)
        err = (out.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
        assert err < 2e-2, err
    else:
        assert_close(out, ref, dt, msg="dx")
    if post:
        v = out.cpu().float().double()
        gq = v * (mask.double()[torch.arange(M) // rpg, None] if post == 2 else 1.0)
        xh = (py.double() - pbn[2].double()) * pbn[3].double()
        st = stats.cpu().sum(0)
        for got, want, nm in ((st[0], gq.sum(0), "sum g"), (st[1], (gq * xh).sum(0), "sum g xhat")):
            e = (got - want).abs().max().item() / max(want.abs().max().item(), 1e-6)
            assert e < (1e-5 if dt == "f32" else 1e-4), f"{nm}: {e:.2e}"


CASES = [  # M, K, N, dyp, store dy, residual, post mode
    (200, 192, 48, True, True, False, 0),
    (130, 384, 96, True, True, True, 1),
    (257, 672, 112, True, True, True, 2),      # K = 10.5 chunks: the ragged last chunk; ragged last row tile
    (64, 1152, 192, True, True, True, 2),
    (100, 576, 192, True, False, False, 1),    # frozen parameters: dy is not stored
    (150, 320, 96, False, False, True, 0),     # materialised dy as the operand
]


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("M,K,N,dyp,store,res,post", CASES)
def test_pw_dgrad(be, dt, M, K, N, dyp, store, res, post):
    _run(be, dt, M, K, N, dyp, store, res, post)


@pytest.mark.gpu
@pytest.mark.parametrize("M,K,N", [(18400, 1152, 192), (73600, 672, 112), (73600, 384, 96), (18400, 576, 192), (294400, 192, 48)])
def test_real_layer_shapes(be_gpu, M, K, N):
    _run(be_gpu, "bf16", M, K, N, True, True, True, 2, seed=7)


def test_bad_arguments_are_errors(be):
    assert be.lib.fn["pw_dgrad_ok"](100, 100, 192) == 0 and be.lib.fn["pw_dgrad_ok"](100, 192, 64) == 0
    with pytest.raises(cabi.MdsError):
        be.call("pw_dgrad", cabi.make("mds_pw_dgrad_args", dtype=1, M=64, K=192, N=64))
