"""Phase timeline of one block of the K-streaming kernel (experiment build -DPWK_TRACE=<block>: libmds_trace.so.bin).
   python tools/pwk_trace.py  -> per phase: mean cycles over the steady-state stages, per wave"""
import os, sys, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ball-action-spotting_amd")]
import torch
from mds import cabi
cabi.HIP_LIB = os.path.join(ROOT, "ball-action-spotting_amd", "csrc", "libmds_trace.so.bin")
lib = cabi.load()
dev = torch.device("cuda:0")


def frag_pack(lib, w):
    """fragment-major bf16 copy of a [N][K] filter (MDS_PACK_FRAG_OI) through mds_pack_weights"""
    N, K = w.shape
    src = w.float().contiguous()
    dst = torch.empty(-(-K // 32) * -(-N // 16) * 512, dtype=torch.bfloat16, device=w.device)
    job = cabi.STRUCTS["mds_pack_job"]()
    job.src, job.dst, job.kind, job.O, job.I, job.taps = src.data_ptr(), dst.data_ptr(), cabi.MDS_PACK_FRAG_OI, N, K, 1
    tab = torch.frombuffer(bytearray(bytes(job)), dtype=torch.uint8).to(w.device)
    lib.check(lib.fn["pack_weights"](tab.data_ptr(), 1, dst.numel(), cabi.MDS_BF16, torch.cuda.current_stream().cuda_stream), "pack_weights")
    torch.cuda.synchronize()
    return dst

BF = torch.bfloat16
def rnd(*s): return torch.randn(*s, device=dev).to(BF)
PH = ["wait_vm+lgkm", "barrier", "issue", "transform", "frag reads", "mfma issue", "(next)"]
for (M, K, N, mode, tag, dx, dw) in [(18400, 1152, 192, 3, "fwd 1152->192", 4, 3), (18400, 1152, 192, 0, "dgrad-like 1152->192 (no prologue)", 4, 3),
                                     (73600, 672, 112, 3, "fwd 672->112", 2, 2)]:
    x = rnd(M, K); w = rnd(N, K); y = torch.empty(M, N, device=dev, dtype=BF)
    sc = torch.rand(K, device=dev) + 0.5; sh = torch.randn(K, device=dev) * 0.1
    gate = torch.rand(20, K, device=dev); st = torch.zeros(32, 2, N, device=dev, dtype=torch.float64)
    trc = torch.zeros(4096, device=dev, dtype=torch.int64)
    lib.fn["dev_set"](18, 2)
    a = cabi.make("mds_pw_fwd_args", dtype=1, M=M, K=K, N=N, x=x, w=w, y=y, pro=cabi.pro(mode, sc, sh, gate, M // 20), residual=None, stats=st,
                  split_part=trc.view(torch.float32), w_frag=frag_pack(lib, w))
    for _ in range(5):
        lib.call("pw_fwd", a, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    S = (K + 63) // 64
    t = trc[: (S + 1) * 32].view(S + 1, 4, 8).cpu().double()
    print(f"== {tag} dx={dx} dw={dw}: block total {(t[S, :, 0] - t[0, :, 0]).mean():.0f} cycles for {S} stages = {(t[S, :, 0] - t[0, :, 0]).mean() / S:.0f} per stage")
    lo, hi = 2, S - 2
    for wv in range(4):
        seg = []
        for ph in range(6):
            seg.append((t[lo:hi, wv, ph + 1] - t[lo:hi, wv, ph]).mean().item())
        seg.append((t[lo + 1:hi + 1, wv, 0] - t[lo:hi, wv, 6]).mean().item())
        print(f"  wave {wv}: " + "  ".join(f"{n} {v:6.0f}" for n, v in zip(["wait", "barrier", "issue", "reads+Wwait", "mfma", "transform", "loop"], seg)))
