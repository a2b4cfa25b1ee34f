"""Developer tool (GPU box): split-K factor sweep of mds_pw_fwd on the inference shapes of one 736x1280 frame."""
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
        a = cabi.make("mds_pw_fwd_args", dtype=0, M=M, K=K, N=N, x=x, w=w, y=y, pro=pro, residual=r, stats=None,
                      epi=cabi.make("mds_epi_t", mode=1, scale=sc, shift=sh), split=S, split_part=part if S else None, split_ticket=tk if S else None)
        out.append(f"S={S}: {t_us(lambda: lib.call('pw_fwd', a, st())):6.1f}")
    # (the ablation columns quoted in k_pw.hip - no partial stores / no reduce loads / no ticket - came from temporary hooks in the kernel)
    print(f"M={M} K={K} N={N} gated={gated}  us per launch\n  " + "\n  ".join(out), flush=True)
