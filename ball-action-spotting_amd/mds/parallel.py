"""Data parallelism for the hot path: one process per MI355X, frame-windows sharded by rank.

The reference is single-GPU: windows are independent samples and BatchNorm is
per-device, so rank r running its own batch of 4 windows with local BN statistics is the very
computation the reference does on one GPU.  The only exchange is a sum-all-reduce of the
parameter gradients.  The engine hands over ONE flat fp32 gradient buffer (6.77 M elements,
27.1 MB) at the end of backward, so the exchange is a single RCCL all-reduce over xGMI — the
largest message a ring can get here, which is what a per-link-bound (7 x ~153 GB/s, no switch)
fabric wants — issued on the compute stream's tail (a 27 MB ring all-reduce is ~0.3 ms vs a
multi-ms step).  No collective touches the data path.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_windows(num_windows: int, rank: int, world: int):
    """Contiguous, balanced assignment of frame-windows (dataset indices) to ranks."""
    base, extra = divmod(num_windows, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def allreduce_mean_(flat: torch.Tensor, group=None, force: bool = False) -> torch.Tensor:
    """In-place mean all-reduce of the flat gradient buffer (backend nccl == RCCL on ROCm).
    `force` issues the collective even in a world of one (single-GPU test of the RCCL call path)."""
    if not dist.is_available() or not dist.is_initialized():
        return flat
    world = dist.get_world_size(group)
    if world == 1 and not force:
        return flat
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.div_(world)
    return flat


def broadcast_state(module: torch.nn.Module, src: int = 0, group=None):
    """Make every rank start from rank `src`'s parameters and buffers (as DDP does at wrap time)."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t, src=src, group=group)


def data_parallel(module, group=None, broadcast: bool = True, force_collective: bool = False):
    """Turn on gradient averaging across ranks for an mds.MultiDimStacker (returns the module).

    Kept as an attribute hook instead of a wrapper class so that ``argus``' attribute access
    (``nn_module.conv2d_encoder`` in src/argus_models.py:108) keeps working."""
    if broadcast:
        broadcast_state(module, 0, group)
    module._grad_sync = lambda flat: allreduce_mean_(flat, group, force_collective)
    return module
