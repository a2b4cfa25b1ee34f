import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ball-action-spotting_amd")]
import torch
from mds import cabi
lib = cabi.load(); dev = torch.device("cuda:0")
M, K, N, S = 300, 448, 144, 3
g = torch.Generator().manual_seed(1)
x = torch.randn(M, K, generator=g).to(dev); w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
part = torch.full((S * M * N,), float("nan"), device=dev)
ticket = torch.zeros(64 * 32, dtype=torch.int32, device=dev)
def run(sp):
    y = torch.full((M, N), float("nan"), device=dev)
    a = cabi.make("mds_pw_fwd_args", dtype=0, M=M, K=K, N=N, x=x, w=w, y=y, pro=cabi.pro(0), residual=None, stats=None,
                  split=sp, split_part=part if sp else None, split_ticket=ticket if sp else None)
    lib.call("pw_fwd", a, torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
    return y
y0 = run(0); y1 = run(S)
ref = x @ w.t()
print("unsplit err", (y0 - ref).abs().max().item(), "split err", (y1 - ref).abs().max().item(), "nan in y1", int(torch.isnan(y1).sum()))
bad = ((y1 - ref).abs() > 1e-3) | torch.isnan(y1)
print("bad per column mod 4:", [int(bad[:, j::4].sum()) for j in range(4)], "bad rows", int(bad.any(1).sum()), "of", M)
p = part.view(S, M, N)
for z in range(S):
    pz = p[z]
    print("z", z, "nan", int(torch.isnan(pz).sum()), "col mod 4 nan:", [int(torch.isnan(pz[:, j::4]).sum()) for j in range(4)])
full = p.sum(0)
print("sum of partials vs ref", (full - ref).abs().max().item(), " y1 vs sum of partials", (y1 - full).abs().max().item())
print("ticket", ticket[:8].tolist())
