"""What does the second stream's hand-off cost the dependent chain?  200 dependent tiny launches (bn_finalize) on the main stream,
(a) alone, (b) with event.record(main) + side.wait_event(event) between them, (c) plus a tiny launch on the side stream after each wait,
(d) events recorded but never waited for."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ball-action-spotting_amd")]
import torch
from mds import cabi
lib = cabi.load(); dev = torch.device("cuda:0")
C = 1152
stats = torch.zeros(cabi.MDS_STAT_SLOTS, 2, C, device=dev, dtype=torch.float64)
gamma = torch.ones(C, device=dev); beta = torch.zeros(C, device=dev); out = torch.empty(4, C, device=dev)
rm = torch.zeros(C, device=dev); rv = torch.ones(C, device=dev); nbt = torch.zeros((), dtype=torch.long, device=dev)
a = cabi.make("mds_bn_finalize_args", C=C, count=1000, stats=stats, gamma=gamma, beta=beta, eps=1e-3, momentum=0.1, training=1,
              running_mean=rm, running_var=rv, num_batches_tracked=nbt, out=out)
out2 = torch.empty(4, C, device=dev)
b = cabi.make("mds_bn_finalize_args", C=C, count=1000, stats=stats, gamma=gamma, beta=beta, eps=1e-3, momentum=0.1, training=0,
              running_mean=rm, running_var=rv, num_batches_tracked=None, out=out2)
side = torch.cuda.Stream()
evs = [torch.cuda.Event() for _ in range(256)]
N = 200


def run(mode):
    main = torch.cuda.current_stream()
    ms, ss = main.cuda_stream, side.cuda_stream
    for k in range(N):
        lib.call("bn_finalize", a, ms)
        if mode >= 1:
            evs[k].record(main)
        if mode in (1, 2):
            side.wait_event(evs[k])
        if mode == 2:
            lib.call("bn_finalize", b, ss)
    if mode in (1, 2):
        main.wait_stream(side)


# the host must be AHEAD of the GPU (as it is inside the training step): a long sleep kernel goes first, everything is enqueued behind it
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
cyc = 10_000_000
e0.record(); torch.cuda._sleep(cyc); e1.record(); torch.cuda.synchronize()
ms_sleep = e0.elapsed_time(e1)
while ms_sleep < 30:
    cyc *= 2
    e0.record(); torch.cuda._sleep(cyc); e1.record(); torch.cuda.synchronize()
    ms_sleep = e0.elapsed_time(e1)
print(f"sleep kernel: {ms_sleep:.1f} ms")
import time
for mode, name in ((0, "alone"), (1, "record + wait"), (2, "record + wait + side launch"), (3, "record only")):
    for _ in range(2):
        run(mode)
    torch.cuda.synchronize()
    torch.cuda._sleep(cyc)
    e0.record()
    t0 = time.perf_counter()
    run(mode)
    t1 = time.perf_counter()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:32s} {e0.elapsed_time(e1) / N * 1e3:7.2f} us per dependent launch on the GPU (host enqueue {(t1 - t0) / N * 1e6:.2f} us each, sleep {ms_sleep:.0f} ms)", flush=True)
