"""Parameter containers of the drop-in MultiDimStacker.

These nn.Modules only *hold* parameters and buffers so that ``state_dict()`` has exactly the
515 keys / shapes / order of the reference model (reference
``src/models/multidim_stacker.py:137-208`` + timm 0.9.2 ``tf_efficientnetv2_b0`` features).
They are never called: all arithmetic runs in the HIP kernels driven by ``engine.py``.
"""
from __future__ import annotations

import math

import torch
from torch import nn

# (type, repeats, kernel, stride, expand, out_chs, se_ratio) — tf_efficientnetv2_b0
ARCH_B0 = [
    ("cn", 1, 3, 1, 1, 16, 0.0),
    ("er", 2, 3, 2, 4, 32, 0.0),
    ("er", 2, 3, 2, 4, 48, 0.0),
    ("ir", 3, 3, 2, 4, 96, 0.25),
    ("ir", 5, 3, 1, 6, 112, 0.25),
    ("ir", 8, 3, 2, 6, 192, 0.25),
]
STEM_CHS = 32
ENC_BN_EPS = 1e-3          # tf_ models: bn_eps = 1e-3
TAIL_BN_EPS = 1e-5         # nn.BatchNorm2d/3d default used by the projections and 3D blocks


class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter container: the computation runs in the mds HIP engine")


def _bn2d(c, eps):
    return nn.BatchNorm2d(c, eps=eps)


class ConvBnActP(_Holder):
    kind = "cn"

    def __init__(self, cin, cout, stride, dpr):
        super().__init__()
        self.cin, self.cout, self.stride, self.dpr = cin, cout, stride, dpr
        self.has_skip = stride == 1 and cin == cout
        self.conv = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = _bn2d(cout, ENC_BN_EPS)


class EdgeResidualP(_Holder):
    kind = "er"

    def __init__(self, cin, cout, stride, exp, dpr):
        super().__init__()
        self.cin, self.cout, self.stride, self.dpr = cin, cout, stride, dpr
        self.mid = cin * exp
        self.has_skip = stride == 1 and cin == cout
        self.conv_exp = nn.Conv2d(cin, self.mid, 3, stride, 1, bias=False)
        self.bn1 = _bn2d(self.mid, ENC_BN_EPS)
        self.conv_pwl = nn.Conv2d(self.mid, cout, 1, bias=False)
        self.bn2 = _bn2d(cout, ENC_BN_EPS)


class SqueezeExciteP(_Holder):
    def __init__(self, chs, rd, conv):
        super().__init__()
        self.rd = rd
        self.conv_reduce = conv(chs, rd, 1, bias=True)
        self.conv_expand = conv(rd, chs, 1, bias=True)


class InvertedResidualP(_Holder):
    kind = "ir"

    def __init__(self, cin, cout, stride, exp, se_ratio, dpr):
        super().__init__()
        self.cin, self.cout, self.stride, self.dpr = cin, cout, stride, dpr
        self.mid = cin * exp
        self.has_skip = stride == 1 and cin == cout
        self.conv_pw = nn.Conv2d(cin, self.mid, 1, bias=False)
        self.bn1 = _bn2d(self.mid, ENC_BN_EPS)
        self.conv_dw = nn.Conv2d(self.mid, self.mid, 3, stride, 1, groups=self.mid, bias=False)
        self.bn2 = _bn2d(self.mid, ENC_BN_EPS)
        self.se = SqueezeExciteP(self.mid, int(round(self.mid * (se_ratio / exp))), nn.Conv2d)
        self.conv_pwl = nn.Conv2d(self.mid, cout, 1, bias=False)
        self.bn3 = _bn2d(cout, ENC_BN_EPS)


def _goog_init(m):
    if isinstance(m, nn.Conv2d):
        fan_out = m.kernel_size[0] * m.kernel_size[1] * m.out_channels // m.groups
        nn.init.normal_(m.weight, 0.0, math.sqrt(2.0 / fan_out))
        if m.bias is not None:
            nn.init.zeros_(m.bias)
    elif isinstance(m, nn.BatchNorm2d):
        nn.init.ones_(m.weight)
        nn.init.zeros_(m.bias)


class EncoderP(_Holder):
    """Parameters of timm's EfficientNetFeatures(tf_efficientnetv2_b0, out_indices=[4])."""

    def __init__(self, in_chans=3, drop_path_rate=0.0):
        super().__init__()
        assert in_chans == 3, "the HIP stem kernel is written for stack_size == 3 frame triples"
        self.conv_stem = nn.Conv2d(in_chans, STEM_CHS, 3, 2, 1, bias=False)
        self.bn1 = _bn2d(STEM_CHS, ENC_BN_EPS)
        total = sum(a[1] for a in ARCH_B0)
        stages, cin, idx = [], STEM_CHS, 0
        for (kind, reps, k, s, e, c, se) in ARCH_B0:
            blocks = []
            for r in range(reps):
                stride = s if r == 0 else 1
                dpr = drop_path_rate * idx / total
                if kind == "cn":
                    blocks.append(ConvBnActP(cin, c, stride, dpr))
                elif kind == "er":
                    blocks.append(EdgeResidualP(cin, c, stride, e, dpr))
                else:
                    blocks.append(InvertedResidualP(cin, c, stride, e, se, dpr))
                cin = c
                idx += 1
            stages.append(nn.Sequential(*blocks))
        self.blocks = nn.Sequential(*stages)
        self.feature_info = [dict(num_chs=c, reduction=r) for c, r in
                             ((16, 2), (32, 4), (48, 8), (112, 16), (192, 32))]
        self.apply(_goog_init)

    def block_list(self):
        return [b for stage in self.blocks for b in stage]


class BatchNormAct3dP(_Holder):
    def __init__(self, c):
        super().__init__()
        self.bn3d = nn.BatchNorm3d(c)


class InvertedResidual3dP(_Holder):
    """Parameters of reference InvertedResidual3d (multidim_stacker.py:93-134)."""

    def __init__(self, cin, cout, expansion_ratio, se_reduce_ratio, drop_path_rate):
        super().__init__()
        self.cin, self.cout, self.dpr = cin, cout, drop_path_rate
        self.mid = cin * expansion_ratio
        self.conv_pw = nn.Conv3d(cin, self.mid, (1, 1, 1), bias=False)
        self.bn1 = BatchNormAct3dP(self.mid)
        self.conv_dw = nn.Conv3d(self.mid, self.mid, (3, 3, 3), padding=(1, 1, 1), groups=self.mid, bias=False)
        self.bn2 = BatchNormAct3dP(self.mid)
        self.se = SqueezeExciteP(self.mid, self.mid // se_reduce_ratio, nn.Conv3d)
        self.conv_pwl = nn.Conv3d(self.mid, cout, (1, 1, 1), bias=False)
        self.bn3 = BatchNormAct3dP(cout)


class GeneralizedMeanPoolingP(_Holder):
    def __init__(self, norm, eps=1e-6):
        super().__init__()
        self.p = nn.Parameter(torch.ones(1) * norm)
        self.output_size = 1
        self.eps = eps
