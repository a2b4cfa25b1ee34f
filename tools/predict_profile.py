"""Developer tool: frame-by-frame StreamPredictor (the reference API) under rocprofv3 / plain timing.
  python tools/predict_profile.py [n_frames] [chunk] [bf16]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ball-action-spotting_amd")]
import torch
import mds
from mds.predict import StreamPredictor
from bench import CONFIG
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cdt = "bf16" if len(sys.argv) > 3 else None
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = mds.MultiDimStacker(**dict(CONFIG, drop_rate=0.0, drop_path_rate=0.0)).to(dev)
for bn in model.modules():
    if isinstance(bn, torch.nn.modules.batchnorm._BatchNorm):
        bn.momentum = 1.0
model.train()
with torch.no_grad():
    model(torch.rand(1, 15, 736, 1280, device=dev))
model.eval(); model.clear_plans()
pool = torch.randint(0, 256, (64, 720, 1280), dtype=torch.uint8, device=dev)
sp = StreamPredictor(model, frame_size=(1280, 736), tta=False, compute_dtype=cdt)
idx = 0
def feed(k):
    global idx
    out = None
    for _ in range(0, k, chunk):
        sel = torch.arange(idx, idx + chunk, device=dev) % 64
        out = sp.predict_batch(pool[sel], idx)[-1][0]
        idx += chunk
    return out
feed(64)
torch.cuda.synchronize(); t0 = time.perf_counter()
feed(n)
torch.cuda.synchronize(); el = time.perf_counter() - t0
print(f"{n / el:.1f} frames/s  ({el / n * 1e3:.3f} ms per frame), chunk {chunk}, dtype {cdt or 'fp32'}")
t0 = time.perf_counter(); feed(n); host = time.perf_counter() - t0; torch.cuda.synchronize()
print(f"host issue time per frame: {host / n * 1e3:.3f} ms")
