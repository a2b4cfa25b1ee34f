"""1x1-conv GEMM kernels (mds_pw_fwd / mds_pw_wgrad) against a torch fp32 reference."""
import pytest
import torch
import torch.nn.functional as F

from backends import be, DT, assert_close  # noqa: F401
from mds import cabi


def _mk_pro(be, mode, K, groups, g):
    scale = be.t(1.0 + 0.2 * torch.randn(K, generator=g))
    shift = be.t(0.3 * torch.randn(K, generator=g))
    gate = be.t(torch.rand(groups, K, generator=g))
    return scale, shift, gate


def _apply_pro(x, mode, scale, shift, gate, rpg):
    if mode == 0:
        return x
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
    st = torch.zeros(cabi.MDS_STAT_SLOTS, 2, N, device=be.device)
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


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("M,K,N,mode", [
    (500, 32, 64, 0),
    (333, 48, 144, 2),
    (700, 112, 32, 3),
    (64, 192, 16, 1),
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
