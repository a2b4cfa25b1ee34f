"""mds_bn_bwd_apply_wg / mds_wg_finish (k_bwg.hip): the BatchNorm-backward apply pass that also accumulates a 1x1 weight
gradient.  Checked against (a) the plain apply kernel's dy and (b) a float64 matmul of the same operands, on the kernel
simulator and on MI355X; reference arithmetic: torch's native_batch_norm_backward + convolution_backward behind
/root/reference/src/models/multidim_stacker.py:124-134."""
import pytest
import torch

from backends import be, be_gpu, DT, assert_close  # noqa: F401
from mds import cabi


def gen(s):
    return torch.Generator().manual_seed(s)


def _lin(C, g):
    A = 0.5 + torch.rand(C, generator=g)
    B = 0.2 * torch.randn(C, generator=g)
    D = 0.1 * torch.randn(C, generator=g)
    return torch.stack([A, B, D]).contiguous()


def _run(be, dt, M, C, K, se, group_rows, seed=0, knob_blocks=0):
    code, tdt = DT[dt]
    g = gen(seed + M + C + K)
    u = torch.randn(M, C, generator=g).to(tdt)
    y = (torch.randn(M, C, generator=g) * 1.5 + 0.3).to(tdt)
    x = torch.randn(M, K, generator=g).to(tdt)
    lin = _lin(C, g)
    scale = 0.5 + torch.rand(C, generator=g); shift = 0.2 * torch.randn(C, generator=g)
    bn = torch.stack([scale, shift, torch.zeros(C), torch.ones(C)]).contiguous()
    groups = M // group_rows if group_rows else 1
    gate = torch.rand(groups, C, generator=g); dpool = 0.05 * torch.randn(groups, C, generator=g)
    if knob_blocks:
        be.lib.check(be.lib.fn["dev_set"](cabi.MDS_KNOB_BWG_BLOCKS, knob_blocks), "dev_set")
    try:
        slabs = be.lib.fn["bn_bwd_apply_wg_slabs"](M, C, K, group_rows, 1 if se else 0, code)
        assert slabs > 0
        dy = torch.full((M, C), float("nan")).to(tdt).to(be.device)
        part = torch.full((slabs, C, K), float("nan"), device=be.device)
        gs = cabi.gsrc(mode=cabi.MDS_G_SE_SILU if se else cabi.MDS_G_PLAIN, u=be.t(u), gate=be.t(gate) if se else None,
                       dpooled=be.t(dpool) if se else None, rows_per_group=group_rows if se else 0)
        be.call("bn_bwd_apply_wg", cabi.make("mds_bn_bwd_apply_wg_args", dtype=code, M=M, C=C, g=gs, y=be.t(y), bn=be.t(bn),
                                             lin=be.t(lin), dy=dy, K=K, x=be.t(x), wide_act=1 if se else 0,
                                             group_rows=group_rows, slabs=slabs, part=part))
        dw = torch.full((C, K) if not se else (K, C), 0.25, device=be.device)
        be.call("wg_finish", cabi.make("mds_wg_finish_args", C=C, K=K, slabs=slabs, transpose=1 if se else 0, part=part, dw=dw))
        be.sync()
    finally:
        if knob_blocks:
            be.lib.check(be.lib.fn["dev_set"](cabi.MDS_KNOB_BWG_BLOCKS, 0), "dev_set")
    # reference in float64 on the same (rounded) operands
    uf, yf, xf = u.double(), y.double(), x.double()
    A, B, D = lin.double()
    if se:
        rows = torch.arange(M) // group_rows
        z = yf * scale.double() + shift.double()
        s = torch.sigmoid(z)
        gg = (uf * gate.double()[rows] + dpool.double()[rows]) * (s * (1 + z * (1 - s)))
        dy_ref = A * gg + B * yf + D
        wide = (z * s * gate.double()[rows]).to(tdt).double()       # the kernel rounds the staged operand to the storage type
        dw_ref = (wide.t() @ xf).t() + 0.25                          # [K][C]
    else:
        dy_ref = A * uf + B * yf + D
        wide = dy.cpu().double()                                     # the staged operand IS the stored dy (same rounding)
        dw_ref = wide.t() @ xf + 0.25                                # [C][K]
    assert_close(dy, dy_ref, dt, msg="dy")
    err = (dw.double().cpu() - dw_ref).abs().max().item() / dw_ref.abs().max().item()
    # (SE, bf16: the staged silu(z)*gate is rounded from the kernel's fp32 value, the reference's from float64 - one bf16 ulp
    #  on the elements that sit at a rounding boundary)
    assert err < (2e-5 if dt == "f32" else (1e-3 if se else 1e-4)), f"weight gradient rel err {err:.2e} (slabs {slabs})"
    return slabs


# (M, C, K, se, group_rows): every (chunk width, narrow width) pair the planner uses, ragged slabs, one slab and many
CASES = [
    (200, 64, 48, False, 0),        # one chunk, 3 full steps + a ragged one
    (520, 192, 96, False, 0),
    (333, 96, 112, False, 0),       # 96-channel chunks: 192 staging threads
    (300, 128, 192, False, 0),
    (3 * 150, 192, 96, True, 150),  # squeeze-excite source: slabs stay inside a group (an image)
    (2 * 200, 96, 112, True, 200),
    (2 * 130, 64, 192, True, 130),
    (2 * 70, 96, 192, True, 70),
]


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("M,C,K,se,group_rows", CASES)
def test_apply_with_weight_gradient(be, dt, M, C, K, se, group_rows):
    _run(be, dt, M, C, K, se, group_rows)


@pytest.mark.parametrize("M,C,K,se,group_rows,blocks", [(1000, 128, 96, False, 0, 24), (4 * 200, 128, 48, True, 200, 64)])
def test_many_slabs_and_pipeline(be, M, C, K, se, group_rows, blocks):
    """more slabs than XCDs (the slab -> XCD mapping, the rounding of the slab count to 8) and slabs long enough for the
    two-register-set loop to run several trips"""
    slabs = _run(be, "bf16", M, C, K, se, group_rows, seed=3, knob_blocks=blocks)
    assert slabs >= 8


@pytest.mark.gpu
@pytest.mark.parametrize("M,C,K,se,group_rows", [(20 * 920, 1152, 192, False, 0), (20 * 920, 1152, 192, True, 920),
                                                  (20 * 3680, 672, 112, False, 0), (20 * 3680, 672, 112, True, 3680),
                                                  (20 * 3680, 384, 96, True, 3680), (4 * 4600, 576, 192, True, 4600)])
def test_real_layer_shapes(be_gpu, M, C, K, se, group_rows):
    _run(be_gpu, "bf16", M, C, K, se, group_rows, seed=5)


def test_bad_arguments_are_errors(be):
    with pytest.raises(cabi.MdsError):
        be.call("bn_bwd_apply_wg", cabi.make("mds_bn_bwd_apply_wg_args", dtype=1, M=64, C=80, K=48))
    assert be.lib.fn["bn_bwd_apply_wg_slabs"](64, 64, 50, 0, 0, 1) == 0
