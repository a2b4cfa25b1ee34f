"""Is the host the bottleneck of the training step? (developer tool)  Adds busy-wait time to the host side of every step:
if ms/step does not move until the added time exceeds GPU time - host time, the GPU never waits for Python."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ball-action-spotting_amd")]
import torch
import bench, mds
from mds.train import FusedAdamW, FocalLoss
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = mds.MultiDimStacker(**bench.CONFIG).to(dev).train()
opt = FusedAdamW(model.parameters(), lr=1e-4)
crit = FocalLoss()
x = torch.rand(4, 15, 736, 1280, device=dev)
target = torch.randint(0, 2, (4, 2), device=dev).float()
def spin(ms):
    t = time.perf_counter() + ms * 1e-3
    while time.perf_counter() < t:
        pass
def step(extra, where):
    opt.zero_grad(set_to_none=True)
    if where == "start": spin(extra)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = crit(model(x), target)
    if where == "mid": spin(extra)
    loss.backward()
    if where == "end": spin(extra)
    opt.step()
for where in ("start", "mid", "end"):
    for extra in (0.0, 0.5, 1.0, 2.0, 3.0, 4.0):
        for _ in range(3): step(extra, where)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): step(extra, where)
        torch.cuda.synchronize()
        print(f"{where:5s} +{extra:.1f} ms of host work per step -> {1e3 * (time.perf_counter() - t0) / 20:.2f} ms/step", flush=True)
