import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ball-action-spotting_amd")]
import torch
from mds import cabi
lib = cabi.load(); dev = torch.device("cuda:0")
def t_us(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
st = lambda: torch.cuda.current_stream().cuda_stream
for (M, K, N, gated) in [(920, 1152, 192, 1), (3680, 672, 112, 1), (920, 192, 1152, 0), (4600, 576, 192, 1)]:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5; y = torch.empty(M, N, device=dev)
    gate = torch.rand(8, K, device=dev); sc = torch.rand(N, device=dev); sh = torch.rand(N, device=dev); r = torch.randn(M, N, device=dev)
    part = torch.empty(16 * M * N, device=dev); tk = torch.zeros(4096 * 32, dtype=torch.int32, device=dev)
    pro = cabi.pro(4, None, None, gate, (M + 7) // 8) if gated else cabi.pro(0)
    out = []
    for S in (0, 2, 3, 4, 6, 8, 12):
        row = []
        for dbg in (0, 1, 2, 3, 7):
            a = cabi.make("mds_pw_fwd_args", dtype=0, M=M, K=K, N=N, x=x, w=w, y=y, pro=pro, residual=r, stats=None,
                          epi=cabi.make("mds_epi_t", mode=1, scale=sc, shift=sh), split=S, split_part=part if S else None, split_ticket=tk if S else None, K1=dbg if S else 0)
            row.append(t_us(lambda: lib.call("pw_fwd", a, st())))
            tk.zero_()
            if not S: break
        out.append(f"S={S}: " + " ".join(f"{v:6.1f}" for v in row))
    print(f"M={M} K={K} N={N} gated={gated}  [full | no stores | no reduce loads | neither | neither+no ticket]\n  " + "\n  ".join(out), flush=True)
