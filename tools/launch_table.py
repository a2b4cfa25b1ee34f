"""Every launch of one profiled step with its achieved algorithmic GB/s, ranked by the time it would
save at 4.5 TB/s (developer tool: finds the outliers)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ball-action-spotting_amd")]
import torch
import bench, mds
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = mds.MultiDimStacker(**bench.CONFIG).to(dev).train()
x = torch.rand(4, 15, 736, 1280, device=dev)
target = torch.randint(0, 2, (4, 2), device=dev).float()
def step():
    model.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = bench.focal_loss(model(x), target)
    loss.backward()
for _ in range(3): step()
plan = next(p for pool in model._cache.plans.values() for p in pool if p.kind == "full" and p.need_grad)
plan.profile = []
step(); torch.cuda.synchronize()
rows = []
for idx, (name, seg, e0, e1, (nbytes, flops)) in enumerate(plan.profile):
    us = e0.elapsed_time(e1) * 1e3
    ideal = nbytes / 4.5e6   # us at 4.5 TB/s
    rows.append((us - ideal, us, name, seg, idx, nbytes / 1e6, nbytes / max(us, 1e-3) / 1e3, flops / max(us, 1e-3) / 1e6))
rows.sort(reverse=True)
print(f"{'excess':>8} {'us':>8}  kernel           seg   idx     MB     GB/s    TF/s")
for r in rows[:int(sys.argv[1]) if len(sys.argv) > 1 else 60]:
    print(f"{r[0]:8.1f} {r[1]:8.1f}  {r[2]:16s} {r[3]:5s} {r[4]:4d} {r[5]:7.1f} {r[6]:8.0f} {r[7]:7.1f}")
