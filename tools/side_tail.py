"""Developer tool: at the end of the backward, which stream finishes last - the dependent chain or the weight-gradient stream - and by how much?
Event pairs recorded on both streams right before Plan.join_backward (untraced run)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ball-action-spotting_amd")]
import torch
import bench, mds
from mds import engine
from mds.train import FusedAdamW, FocalLoss
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = mds.MultiDimStacker(**bench.CONFIG).to(dev).train()
opt = FusedAdamW(model.parameters(), lr=1e-4)
crit = FocalLoss()
x = torch.rand(4, 15, 736, 1280, device=dev)
target = torch.randint(0, 2, (4, 2), device=dev).float()
recs = []
orig = engine.Plan.join_backward
def join(self):
    side = getattr(self, "_side", None)
    if side is not None:
        em, es, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), self._t0
        em.record(torch.cuda.current_stream(self.device)); es.record(side)
        recs.append((e0, em, es))
    orig(self)
engine.Plan.join_backward = join
orig_bb = engine.Plan.begin_backward
def bb(self):
    self._t0 = torch.cuda.Event(enable_timing=True); self._t0.record()
    orig_bb(self)
engine.Plan.begin_backward = bb
def step():
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = crit(model(x), target)
    loss.backward()
    opt.step()
for _ in range(5): step()
torch.cuda.synchronize(); recs.clear()
for _ in range(10): step()
torch.cuda.synchronize()
cm = sum(e0.elapsed_time(em) for e0, em, es in recs) / len(recs)
cs = sum(e0.elapsed_time(es) for e0, em, es in recs) / len(recs)
print(f"backward start -> chain's last kernel done {cm:.3f} ms; -> weight-gradient stream drained {cs:.3f} ms; the step waits {max(cs - cm, 0):.3f} ms for the second stream")
