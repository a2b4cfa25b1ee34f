"""Data parallelism for the hot path: one process per MI355X, frame-windows sharded by rank.

The reference is single-GPU: windows are independent samples and BatchNorm is
per-device, so rank r running its own batch of 4 windows with local BN statistics is the very
computation the reference does on one GPU.  The only exchange is a sum-all-reduce of the
parameter gradients.  The engine hands over ONE flat fp32 gradient buffer (6.77 M elements,
27.1 MB) at the end of backward, so the exchange is a single RCCL all-reduce over xGMI — the
largest message a ring can get here, which is what a per-link-bound (7 x ~153 GB/s, no switch)
fabric wants — issued on the compute stream's tail (a 27 MB ring all-reduce is ~0.3 ms vs a
multi-ms step).  No collective touches the data path.  `BucketedSync` (the default) splits that buffer into a few
slices in backward order and overlaps their all-reduces with the rest of the backward pass.
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
    """Make every rank start from rank `src`'s parameters and buffers (as DDP does at wrap time): ONE broadcast of a packed
    fp32 buffer (all 6.77 M parameters + the BatchNorm running statistics, ~27.5 MB - a single large message instead of
    ~830 small ones on a point-to-point fabric) plus one of the packed integer buffers (num_batches_tracked).
    BatchNorm buffers are NOT re-synchronised afterwards: every rank keeps the running statistics of its own shards,
    exactly what the reference's DataParallel / single-GPU runs and torch DDP (broadcast_buffers aside) leave to rank 0's
    checkpoint - the EMA copy that is validated and saved lives on rank 0."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    with torch.no_grad():
        tensors = list(module.parameters()) + list(module.buffers())
        for floating in (True, False):
            ts = [t for t in tensors if t.is_floating_point() == floating]
            if not ts:
                continue
            dt = torch.float32 if floating else torch.int64
            flat = torch.cat([t.detach().reshape(-1).to(dt) for t in ts])
            dist.broadcast(flat, src=src, group=group)
            off = 0
            for t in ts:
                n = t.numel()
                t.copy_(flat[off:off + n].view(t.shape).to(t.dtype))
                off += n


XGMI_LINK_GBS = 153.0       # one xGMI link, one direction (MI355X_MICROARCH.md: 7 links per GPU, point-to-point)
RING_HOP_US = 6.0           # per ring step: RCCL's own kernel hand-off on a link (latency term of a small slice)
RCCL_CU_TAX = 0.25          # share of an all-reduce's duration its kernels take from the overlapped backward (measured for the
                            # weight-gradient stream, which competes for CUs the same way: DESIGN 0e)


def ring_allreduce_ms(nbytes: float, world: int) -> float:
    """Ring all-reduce of `nbytes` over point-to-point xGMI: 2 (N - 1) steps, each moving nbytes / N over ONE link."""
    if world <= 1:
        return 0.0
    steps = 2 * (world - 1)
    return steps * (nbytes / world / (XGMI_LINK_GBS * 1e9) * 1e3 + RING_HOP_US * 1e-3)


def predict_step_ms(local_ms: float, world: int, slice_bytes) -> dict:
    """The stated scaling model, so that a measured N-GPU line judges itself: per-rank work is fixed (weak scaling), the
    gradient arena goes out in `slice_bytes` slices in backward order from a communication stream; every slice but the LAST is
    hidden behind the rest of the backward, the last one is exposed, and RCCL's kernels tax the overlapped backward by RCCL_CU_TAX
    of their duration.  local_ms: the step without any exchange (the N = 1 time on the same device)."""
    slice_bytes = list(slice_bytes)
    total = ring_allreduce_ms(sum(slice_bytes), world) if slice_bytes else 0.0
    exposed = ring_allreduce_ms(slice_bytes[-1], world) if slice_bytes else 0.0
    hidden = max(sum(ring_allreduce_ms(b, world) for b in slice_bytes[:-1]), 0.0)
    pred = local_ms + exposed + RCCL_CU_TAX * hidden
    return {"predicted_ms": round(pred, 4), "predicted_exposed_ms": round(exposed, 4), "predicted_allreduce_total_ms": round(total, 4),
            "predicted_efficiency": round(local_ms / pred, 4) if pred > 0 else None,
            "model": f"ring all-reduce over xGMI at {XGMI_LINK_GBS:.0f} GB/s per link + {RING_HOP_US:.0f} us per ring step; last slice exposed, "
                     f"{RCCL_CU_TAX:.2f} of the hidden slices' duration taxed on the backward"}


class BucketedSync:
    """Gradient averaging overlapped with the backward pass.

    The engine reports, while it is still issuing launches, that a slice arena[lo:hi] of the flat gradient buffer is
    final (every kernel that writes it has been enqueued — on the main stream or on the weight-gradient stream): the
    slice is all-reduced in place from a communication stream that waits for exactly those two points, in the order
    head -> 3D tail -> stage 5 ... stem.  Slices are >= 1.5 M elements (6 MB): xGMI is point-to-point, a ring
    all-reduce is bound per link, so few large messages beat many small ones.  `finish` makes the compute stream
    wait for the collectives before the gradients are handed to autograd."""

    def __init__(self, group=None, force: bool = False):
        self.group, self.force = group, force
        self.handles, self.comm, self.events = [], None, []
        self.paused = False          # bench: steps without the exchange = the N = 1 time on this very device (local gradients only)
        self.timing = False          # bench: record an event pair around every slice's all-reduce on the communication stream
        self.last_timing = None      # [(lo, hi, microseconds)] of the last step when `timing` is on
        self.last_exposed_ms = None  # with `timing`: end of the last backward kernel -> end of the last all-reduce (what the step waits for)

    def active(self):
        return (not self.paused) and dist.is_available() and dist.is_initialized() and (dist.get_world_size(self.group) > 1 or self.force)

    def on_cut(self, plan, lo, hi):
        if not self.active():
            return
        buf = plan.grad_arena.tensor[lo:hi]
        if buf.is_cuda:
            if self.comm is None:
                self.comm = torch.cuda.Stream(device=buf.device)
            main = torch.cuda.current_stream(buf.device)
            for s_ in (main, getattr(plan, "_side", None)):
                if s_ is not None:
                    ev = torch.cuda.Event()
                    ev.record(s_)
                    self.comm.wait_event(ev)
                    self.events.append(ev)
            with torch.cuda.stream(self.comm):
                if self.timing:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(self.comm)
                h = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                if self.timing:
                    e1.record(self.comm)
                    self._tev = getattr(self, "_tev", []) + [(lo, hi, e0, e1)]
                self.handles.append((h, lo, hi))
        else:
            self.handles.append((dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True), lo, hi))

    def finish(self, plan):
        """-> world size the sums have to be divided by (1 when nothing was exchanged)"""
        if not self.handles:
            return 1
        for h, _, _ in self.handles:
            h.wait()
        if self.comm is not None:
            cur = torch.cuda.current_stream(plan.device)
            if self.timing:          # the backward's own work ends here on the compute stream; the collectives end on the communication stream
                eb, ec = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                eb.record(cur); ec.record(self.comm)
                self._texp = (eb, ec)
            cur.wait_stream(self.comm)
        self.buckets = [(lo, hi) for _, lo, hi in self.handles]     # kept for inspection by tests
        if self.timing and getattr(self, "_tev", None):
            torch.cuda.synchronize(plan.device)
            self.last_timing = [(lo, hi, round(e0.elapsed_time(e1) * 1e3, 1)) for lo, hi, e0, e1 in self._tev]
            self._tev = []
            if getattr(self, "_texp", None):
                self.last_exposed_ms = round(max(self._texp[0].elapsed_time(self._texp[1]), 0.0), 4)
                self._texp = None
        self.handles, self.events = [], []
        return dist.get_world_size(self.group)

    def __call__(self, flat):      # the unbucketed interface (one all-reduce of the finished buffer)
        return allreduce_mean_(flat, self.group, self.force)


def data_parallel(module, group=None, broadcast: bool = True, force_collective: bool = False, bucketed: bool = True):
    """Turn on gradient averaging across ranks for an mds.MultiDimStacker (returns the module).

    Kept as an attribute hook instead of a wrapper class so that ``argus``' attribute access
    (``nn_module.conv2d_encoder`` in src/argus_models.py:108) keeps working."""
    if broadcast:
        broadcast_state(module, 0, group)
    module._grad_sync = BucketedSync(group, force_collective) if bucketed else (lambda flat: allreduce_mean_(flat, group, force_collective))
    return module
