"""Developer tool: ablation / grid sweep of mds_pw_wgrad at the real layer shapes (MDS_KNOB_WG_DBG, MDS_KNOB_WG_BLOCKS)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ball-action-spotting_amd")]
import torch
from mds import cabi
dev = torch.device("cuda:0"); lib = cabi.load(); BF = torch.bfloat16
SHAPES = [(18400, 192, 1152, 0, "b5 pw 192->1152"), (18400, 1152, 192, 4, "b5 pwl 1152->192 gate"), (73600, 112, 672, 0, "b4 pw"),
          (73600, 672, 112, 4, "b4 pwl gate"), (18400, 192, 576, 0, "3d pw"), (73600, 96, 384, 0, "b3 pw"), (294400, 48, 192, 0, "b3.0 pw")]
def t_us(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (M, K, N, mode, tag) in SHAPES:
    x = torch.randn(M, K, device=dev).to(BF); dy = torch.randn(M, N, device=dev).to(BF); dw = torch.zeros(N, K, device=dev)
    gate = torch.rand(20, K, device=dev)
    a = cabi.make("mds_pw_wgrad_args", dtype=1, M=M, K=K, N=N, x=x, dy=dy, dw=dw, pro=cabi.pro(mode, None, None, gate, M // 20))
    s = torch.cuda.current_stream().cuda_stream
    row = []
    for dbg, blocks in [(64, 0), (0, 0), (1, 0), (4, 0), (0, 64), (0, 128), (0, 192), (0, 384), (0, 512), (0, 768)]:
        lib.check(lib.fn["dev_set"](4, dbg), "dev_set"); lib.check(lib.fn["dev_set"](5, blocks), "dev_set")
        row.append(f"d{dbg}b{blocks}:{t_us(lambda: lib.call('pw_wgrad', a, s)):6.1f}")
    print(f"{tag:26s} MB={(M*K+M*N)*2/1e6:6.1f}  " + "  ".join(row), flush=True)
