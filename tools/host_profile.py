"""cProfile of the host side of the training step (developer tool): where do the 11 ms of issue time go?"""
import cProfile, os, pstats, sys, time
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
pr = cProfile.Profile()
pr.enable()
for _ in range(20): step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("cumulative")
st.print_stats(28)
