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
for (M, K, N, mode, tag) in [(18400, 1152, 192, 3, "fwd 1152->192 BN+SiLU+gate"), (18400, 1152, 192, 2, "fwd 1152->192 BN+SiLU"),
                             (18400, 1152, 192, 1, "fwd 1152->192 affine only"), (18400, 1152, 192, 4, "fwd 1152->192 gate only"),
                             (18400, 1152, 192, 0, "no prologue 1152->192"),
                             (73600, 672, 112, 3, "fwd 672->112"), (73600, 672, 112, 0, "no prologue 672->112")]:
    x = rnd(M, K); w = rnd(N, K); y = torch.empty(M, N, device=dev, dtype=BF)
    sc = torch.rand(K, device=dev) + 0.5; sh = torch.randn(K, device=dev) * 0.1
    gate = torch.rand(20, K, device=dev); st = torch.zeros(32, 2, N, device=dev, dtype=torch.float64)
    trc = torch.zeros(8192, device=dev, dtype=torch.int64)
    lib.fn["dev_set"](18, 2)
    a = cabi.make("mds_pw_fwd_args", dtype=1, M=M, K=K, N=N, x=x, w=w, y=y, pro=cabi.pro(mode, sc, sh, gate, M // 20), residual=None, stats=st,
                  split_part=trc.view(torch.float32), w_frag=frag_pack(lib, w))
    for _ in range(5):
        lib.call("pw_fwd", a, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    S = (K + 63) // 64
    NEED = 2 if mode else 1
    T = S + NEED
    t = trc[: (T + 1) * 64].view(T + 1, 8, 8).cpu().double()
    t0 = t[0, 4, 0]
    print(f"== {tag}: first producer step -> end of the K loop {(t[T, :4, 0].max() - t0):.0f} cycles, {S} stages = {(t[T, :4, 0].max() - t0) / S:.0f} per stage; "
          f"loop end -> trace dump {(t[T + 0, 4, 0] - t0):.0f}")
    lo, hi = NEED + 3, T - 3
    for wv in range(4, 8):
        seg = [(t[lo:hi, wv, ph + 1] - t[lo:hi, wv, ph]).mean().item() for ph in range(4)]
        seg.append((t[lo + 1:hi + 1, wv, 0] - t[lo:hi, wv, 4]).mean().item())
        print(f"  producer {wv - 4}: " + "  ".join(f"{n} {v:6.0f}" for n, v in zip(["wait_vm", "barrier", "issue", "transform", "loop"], seg)))
    for wv in range(4):
        seg = [(t[lo:hi, wv, ph + 1] - t[lo:hi, wv, ph]).mean().item() for ph in range(5)]
        seg.append((t[lo + 1:hi + 1, wv, 0] - t[lo:hi, wv, 5]).mean().item())
        print(f"  consumer {wv}: " + "  ".join(f"{n} {v:6.0f}" for n, v in zip(["wait0", "mfma0", "barrier", "wait1", "mfma1", "loop"], seg)))
