"""Deterministic, init-order-independent parameter fill shared by the golden generator and tests.

Every state_dict entry i gets values from ``torch.Generator().manual_seed(seed*100003+i)`` so
the reference classes and the restatement hold identical weights regardless of how their
constructors consume the global RNG.
"""
import torch
from torch import nn


def fill_deterministic(module: nn.Module, seed: int = 0, scale: float = 0.15) -> nn.Module:
    with torch.no_grad():
        for i, (name, t) in enumerate(module.state_dict().items()):
            g = torch.Generator().manual_seed(seed * 100003 + i)
            if not t.is_floating_point():
                t.zero_()
                continue
            leaf = name.split(".")[-1]
            if leaf == "running_var":
                t.copy_(0.5 + torch.rand(t.shape, generator=g))
            elif leaf == "running_mean":
                t.copy_(0.1 * torch.randn(t.shape, generator=g))
            elif leaf == "p":                      # GeM exponent stays near its init
                t.fill_(3.0)
            elif t.ndim == 1 and leaf == "weight":  # norm scale
                t.copy_(1.0 + 0.1 * torch.randn(t.shape, generator=g))
            elif t.ndim == 1:                      # biases
                t.copy_(0.1 * torch.randn(t.shape, generator=g))
            else:
                fan_in = t[0].numel()
                t.copy_(torch.randn(t.shape, generator=g) * (scale + 1.0 / fan_in ** 0.5))
    return module


class FakeEncoder(nn.Module):
    """Tiny stand-in encoder used ONLY to pin forward_2d's frame->channel grouping:
    32x32 average pooling of each input plane, tiled to 192 channels (channel c carries
    plane c % 3 scaled by 1 + c // 3)."""

    def __init__(self, in_chans=3, drop_path_rate=0.0, out_indices=(4,)):
        super().__init__()
        self.feature_info = [dict(num_chs=192)] * 5
        self.in_chans = in_chans
        self.dummy = nn.Parameter(torch.zeros(1))

    def forward(self, x):
        p = torch.nn.functional.avg_pool2d(x, 32)
        reps = 192 // self.in_chans
        scale = (1 + torch.arange(reps, dtype=x.dtype)).repeat_interleave(self.in_chans)
        y = p.repeat(1, reps, 1, 1) * scale.view(1, -1, 1, 1) + self.dummy
        return [y]
