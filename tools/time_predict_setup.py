import sys, os, time
ROOT = "/root/repo" if os.path.exists("/root/repo/bench.py") else os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path[:0] = [ROOT, os.path.join(ROOT, "ball-action-spotting_amd")]
import torch, mds
from mds.predict import StreamPredictor
import bench
dev = torch.device("cuda:0")
model = mds.MultiDimStacker(**dict(bench.CONFIG, drop_rate=0.0, drop_path_rate=0.0)).to(dev).eval()
pool = torch.randint(0, 256, (64, 720, 1280), dtype=torch.uint8, device=dev)
for chunk, lanes in ((8, 0), (1, 0), (8, 3), (1, 4), (1, 4)):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sp = StreamPredictor(model, frame_size=(1280, 736))
    idx = 0
    if lanes:
        for _ in sp.predict_stream((pool[j % 64] for j in range(88)), 0, chunk=chunk, lanes=lanes): pass
    else:
        for j in range(0, 88, chunk):
            sp.predict_batch(pool[j % 64:j % 64 + chunk] if j % 64 + chunk <= 64 else pool[:chunk], j)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    sp.close()
    print(f"chunk={chunk} lanes={lanes}: build + 88 warm-up frames {t1 - t0:.1f} s", flush=True)
