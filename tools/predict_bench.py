"""Sliding-window inference in the access pattern of the reference's MultiDimStackerPredictor
(src/predictors.py:50-72): per new frame one forward_2d on the newest stack of 3 frames (x2 with TTA),
then forward_3d + forward_head on the 5 cached stack features.  Synthetic frames; starting point for
SURVEY §8(f) N1 (developer tool, not the round's metric)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ball-action-spotting_amd")]
import torch
import bench, mds
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = mds.MultiDimStacker(**bench.CONFIG).to(dev).eval()
for tta in (False, True):
    b = 2 if tta else 1
    feats = [torch.randn(b, 1, 192, 23, 40, device=dev) for _ in range(5)]
    frames = torch.rand(b, 3, 736, 1280, device=dev)
    def one():
        with torch.no_grad():
            f = model.forward_2d(frames)
            feats.pop(0); feats.append(f)
            x = model.forward_3d(torch.cat(feats, dim=1))
            return model.forward_head(x)
    for _ in range(5): one()
    torch.cuda.synchronize()
    n = 50
    t0 = time.perf_counter()
    for _ in range(n): out = one()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"tta={tta}: {dt*1e3:.2f} ms per new frame (one new stack per frame, 4 of 5 cached) -> {1/dt:.0f} frames/s, logits {tuple(out.shape)}")
