"""Developer probe (GPU box, under rocprofv3 --kernel-trace --stats): the predictor in chunks of CHUNK frames on one stream."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ball-action-spotting_amd")]
import torch, mds
from mds.predict import StreamPredictor
import bench
dev = torch.device("cuda:0")
model = mds.MultiDimStacker(**dict(bench.CONFIG, drop_rate=0.0, drop_path_rate=0.0)).to(dev)
for bn in model.modules():
    if isinstance(bn, torch.nn.modules.batchnorm._BatchNorm):
        bn.momentum = 1.0
model.train()
with torch.no_grad():
    model(torch.rand(1, 15, 736, 1280, device=dev))
model.eval(); model.clear_plans()
pool = torch.randint(0, 256, (64, 720, 1280), dtype=torch.uint8, device=dev)
chunk = int(os.environ.get("CHUNK", "8"))
sp = StreamPredictor(model, frame_size=(1280, 736), use_graphs=False)
for j in range(0, 64 + 40 * chunk, chunk):
    a = j % 64
    sp.predict_batch(pool[a:a + chunk] if a + chunk <= 64 else pool[:chunk], j)
torch.cuda.synchronize()
