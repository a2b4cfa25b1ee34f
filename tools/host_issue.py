"""Host time to ISSUE one training step vs the GPU time it takes (developer tool): is the launch loop the bottleneck?"""
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
def step():
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = crit(model(x), target)
    loss.backward()
    opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host issue {1e3 * (t1 - t0) / n:.2f} ms/step; wall incl. GPU drain {1e3 * (t2 - t0) / n:.2f} ms/step")
# forward-only and backward-only issue time
torch.cuda.synchronize(); t0 = time.perf_counter()
with torch.autocast("cuda", dtype=torch.bfloat16):
    loss = crit(model(x), target)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"forward: issue {1e3 * (t1 - t0):.2f} ms, done {1e3 * (t2 - t0):.2f} ms")
t0 = time.perf_counter(); loss.backward(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"backward: issue {1e3 * (t1 - t0):.2f} ms, done {1e3 * (t2 - t0):.2f} ms")
