"""Developer probe (GPU box): host enqueue time vs GPU time per frame of StreamPredictor, sequential and pipelined."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ball-action-spotting_amd")]
import torch, mds
from mds.predict import StreamPredictor
import bench
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = mds.MultiDimStacker(**dict(bench.CONFIG, drop_rate=0.0, drop_path_rate=0.0)).to(dev)
for bn in model.modules():
    if isinstance(bn, torch.nn.modules.batchnorm._BatchNorm):
        bn.momentum = 1.0
model.train()
with torch.no_grad():
    model(torch.rand(1, 15, 736, 1280, device=dev))
model.eval(); model.clear_plans()
pool = torch.randint(0, 256, (64, 720, 1280), dtype=torch.uint8, device=dev)
K = int(os.environ.get("K", "300"))
chunk = int(os.environ.get("CHUNK", "1"))
for lanes in [int(v) for v in os.environ.get("LANES", "1,2,3,4,6,8").split(",")]:
    sp = StreamPredictor(model, frame_size=(1280, 736), use_graphs=True)
    idx = 0
    def feed(n):
        global idx
        if lanes:
            for out, _ in sp.predict_stream((pool[(idx + j) % 64] for j in range(n)), idx, chunk=chunk, lanes=lanes):
                pass
        else:
            for j in range(0, n, chunk):
                sp.predict_batch(pool[[(idx + j + k) % 64 for k in range(chunk)]], idx + j)
        idx += n
    feed(64)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); feed(K); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"chunk={chunk} lanes={lanes}: host enqueue {1e6 * (t1 - t0) / K:.0f} us/frame, total {1e6 * (t2 - t0) / K:.0f} us/frame = {K / (t2 - t0):.0f} frames/s", flush=True)
    sp.close()
