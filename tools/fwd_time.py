"""Developer tool (GPU): time of the training FORWARD alone (config 2, bf16, batch 4) - the forward has nothing beside it, so a
kernel's isolated gain there is the step's gain; the knobs were swept inside the whole step, where backward effects dominate.
   MDS_KNOBS="12=3072" python tools/fwd_time.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ball-action-spotting_amd")]
import torch
import bench, mds
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = mds.MultiDimStacker(**bench.CONFIG).to(dev).train()
x = torch.rand(4, 15, 736, 1280, device=dev)
def fwd():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        return model(x)
for _ in range(5):
    out = fwd(); out.sum().backward()          # (a full step once in a while keeps the plan in its steady state)
torch.cuda.synchronize()
ts = []
for rep in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        out = fwd()
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 10 * 1e3)
    out.sum().backward()
print(f"forward ms: best {min(ts):.3f} median {sorted(ts)[2]:.3f}  KNOBS='{os.environ.get('MDS_KNOBS', '')}'")
