"""Experiment: whole-step hipGraph capture through torch.cuda.graph (developer tool)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ball-action-spotting_amd")]
import torch
import bench, mds
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = mds.MultiDimStacker(**bench.CONFIG).to(dev).train()
opt = torch.optim.AdamW(model.parameters(), lr=3e-4, fused=True, capturable=True)
x = torch.rand(4, 15, 736, 1280, device=dev)
target = torch.randint(0, 2, (4, 2), device=dev).float()
def step():
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = bench.focal_loss(model(x), target)
    loss.backward(); opt.step()
    return loss
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for _ in range(n): step()
torch.cuda.synchronize()
print(f"eager  {1e3*(time.perf_counter()-t0)/n:.2f} ms/step", flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    loss = step()
torch.cuda.synchronize()
for _ in range(2): g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n): g.replay()
torch.cuda.synchronize()
print(f"graph  {1e3*(time.perf_counter()-t0)/n:.2f} ms/step  loss {loss.item():.5f}", flush=True)
