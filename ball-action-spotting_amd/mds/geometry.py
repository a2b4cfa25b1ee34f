"""Convolution geometry shared by the module planner and the tests: TF-'SAME' padding
(timm Conv2dSame) and the tap lists consumed by mds_conv_fwd / mds_conv_wgrad."""
import math


def same_pad(i: int, stride: int, k: int = 3):
    """(pad_before, pad_after) of timm's pad_same for input extent i."""
    pad = max((math.ceil(i / stride) - 1) * stride + (k - 1) + 1 - i, 0)
    return pad // 2, pad - pad // 2


def conv_geometry(ih: int, iw: int, stride: int):
    """-> (OH, OW, pad_t, pad_l) for a 3x3 conv: stride 1 uses symmetric pad 1, stride 2 TF-SAME."""
    if stride == 1:
        return ih, iw, 1, 1
    pt, _ = same_pad(ih, stride)
    pl, _ = same_pad(iw, stride)
    return math.ceil(ih / stride), math.ceil(iw / stride), pt, pl


def taps_fwd(pad_t: int, pad_l: int):
    """tap list of the forward conv / of its weight gradient: input = out*stride + (ky - pad)."""
    dy, dx, wi = [], [], []
    for ky in range(3):
        for kx in range(3):
            dy.append(ky - pad_t); dx.append(kx - pad_l); wi.append(ky * 3 + kx)
    return dy, dx, wi


def taps_dgrad_s1():
    """stride-1 data gradient = conv of dy with the flipped/transposed pack (MDS_PACK_IO_FLIP)."""
    return taps_fwd(1, 1)


def taps_dgrad_s2(py: int, px: int, pad_t: int, pad_l: int):
    """stride-2 data gradient for the input-gradient pixels of parity (py, px): y = 2a + py reads
    dy[a + (py + pad_t - ky)/2] for the ky that make it an integer.  Weight index is into the
    flipped pack: t' = 8 - (ky*3 + kx)."""
    dy, dx, wi = [], [], []
    for ky in range(3):
        if (py + pad_t - ky) % 2:
            continue
        for kx in range(3):
            if (px + pad_l - kx) % 2:
                continue
            dy.append((py + pad_t - ky) // 2); dx.append((px + pad_l - kx) // 2)
            wi.append(8 - (ky * 3 + kx))
    return dy, dx, wi
