"""Phase timeline of one block of k_c3.hip (experiment build: make -C ball-action-spotting_amd/csrc c3trace).
   python tools/c3_trace.py  -> per batch of three rows: producers (wait_vm | barrier | issue) and consumers (barrier | compute)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ball-action-spotting_amd")]
import torch
from mds import cabi, geometry as geo
cabi.HIP_LIB = os.path.join(ROOT, "ball-action-spotting_amd", "csrc", "libmds_c3trace.so.bin")
lib = cabi.load()
dev = torch.device("cuda:0")
BF = torch.bfloat16
for (N, H, W, Cin, Cout, res, stats, tag) in [(20, 184, 320, 32, 128, False, True, "b1.1 fwd 32->128 +stats"), (20, 184, 320, 128, 32, True, False, "b1.1 dgrad 128->32 +res")]:
    x = torch.randn(N * H * W, Cin, device=dev).to(BF); w = torch.randn(Cout * 9 * Cin, device=dev).to(BF)
    y = torch.empty(N * H * W, Cout, device=dev, dtype=BF)
    r = torch.randn(N * H * W, Cout, device=dev).to(BF) if res else None
    st = torch.zeros(cabi.MDS_STAT_SLOTS, 2, Cout, device=dev, dtype=torch.float64) if stats else None
    trc = torch.zeros(160 * 32, device=dev, dtype=torch.int64)
    dy, dx, wi = geo.taps_fwd(1, 1)
    a = cabi.make("mds_conv_fwd_args", dtype=1, N=N, IH=H, IW=W, Cin=Cin, OH=H, OW=W, Cout=Cout, A=H, B=W, oy0=0, ox0=0, os=1, **{"is": 1},
                  ntaps=9, dy=dy, dx=dx, wi=wi, wtaps=9, x=x, w=w, y=y, pro=cabi.pro(0), residual=r, stats=st,
                  epi=cabi.make("mds_epi_t", mode=0, scale=trc.view(torch.float32), shift=None))
    for _ in range(4):
        lib.call("conv_fwd", a, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    t = trc.view(160, 8, 4).cpu().double()
    nb = 1                      # the LDS trace area is not cleared: valid batches are the monotonic prefix of consumer 0's stamps
    while nb < 159 and 0 < t[nb, 0, 0] - t[nb - 1, 0, 2] < 1e6 and t[nb, 0, 2] > t[nb, 0, 0]:
        nb += 1
    t0 = t[0, :4, 0].min()
    print(f"== {tag}: {nb} batches in block 100; first stamp -> last {(t[:nb].max() - t0):.0f} cycles = {(t[:nb].max() - t0) / nb:.0f} per batch")
    lo, hi = 2, max(nb - 2, 3)
    for wv in range(4):
        bar = (t[lo:hi, wv, 1] - t[lo:hi, wv, 0]).mean().item(); comp = (t[lo:hi, wv, 2] - t[lo:hi, wv, 1]).mean().item()
        print(f"  consumer {wv}: barrier wait {bar:7.0f}   compute+epilogue {comp:7.0f}")
    for wv in range(4, 8):
        if not (0 < t[lo, wv, 1] - t[lo, wv, 0] < 1e6):
            continue
        seg = [(t[lo:hi, wv, ph + 1] - t[lo:hi, wv, ph]).mean().item() for ph in range(3)]
        loop = (t[lo + 1:hi + 1, wv, 0] - t[lo:hi, wv, 3]).mean().item()
        print(f"  producer {wv - 4}: wait_vm {seg[0]:7.0f}   barrier {seg[1]:7.0f}   issue {seg[2]:7.0f}   loop {loop:6.0f}")
    print("  first 6 batches, consumer 0 (barrier, compute):", [(int(t[b, 0, 1] - t[b, 0, 0]), int(t[b, 0, 2] - t[b, 0, 1])) for b in range(min(6, nb))])
