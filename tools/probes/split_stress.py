"""Developer tool (GPU box): split-K hand-off under load - every launch must reproduce the first one bit for bit while a second stream
streams 1 GB copies through HBM / the L2s (timing perturbation); the partial buffer is poisoned between launches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ball-action-spotting_amd")]
import torch
from mds import cabi
lib = cabi.load(); dev = torch.device("cuda:0")
side = torch.cuda.Stream()
big = torch.empty(256 << 20, dtype=torch.float32, device=dev); big2 = torch.empty_like(big)
bad = 0
for (M, K, N, S, dt) in [(920, 1152, 192, 12, 0), (3680, 672, 112, 4, 0), (920, 1152, 192, 6, 1), (300, 448, 144, 3, 0), (4600, 576, 192, 4, 1)]:
    tdt = torch.bfloat16 if dt else torch.float32
    x = torch.randn(M, K, device=dev).to(tdt); w = (torch.randn(N, K, device=dev) / K ** 0.5).to(tdt); y = torch.empty(M, N, device=dev, dtype=tdt)
    part = torch.empty(S * M * N, device=dev); tk = torch.zeros(4096 * 32, dtype=torch.int32, device=dev)
    a = cabi.make("mds_pw_fwd_args", dtype=dt, M=M, K=K, N=N, x=x, w=w, y=y, pro=cabi.pro(0), residual=None, stats=None, split=S, split_part=part, split_ticket=tk)
    st = torch.cuda.current_stream().cuda_stream
    lib.call("pw_fwd", a, st); torch.cuda.synchronize(); ref = y.clone()
    n = 1500
    for it in range(n):
        if it % 50 == 0:
            with torch.cuda.stream(side):
                big2.copy_(big)
        part.fill_(float("nan"))
        lib.call("pw_fwd", a, st)
        if not torch.equal(y, ref):
            bad += 1
    torch.cuda.synchronize()
    print(f"M={M} K={K} N={N} split={S} dtype={dt}: {n} launches, mismatches so far {bad}, tickets {int(tk.abs().sum())}", flush=True)
print("STRESS", "OK" if bad == 0 else "FAILED")
