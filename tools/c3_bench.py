"""Developer tool: the stride-1 3x3 layers through k_c3.hip alone, with its ablation bits (MDS_KNOB_C3_DBG) - which phase binds?
   python tools/c3_bench.py [dbg values ...]        (default: 0 1 2 4 8 3 6 7 15)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ball-action-spotting_amd")]
import torch
from mds import cabi, geometry as geo
dev = torch.device("cuda:0")
if os.environ.get("C3_LIB"):      # an experiment build (make c3abl ABL=n)
    cabi.HIP_LIB = os.path.join(ROOT, "ball-action-spotting_amd", "csrc", os.environ["C3_LIB"])
lib = cabi.load()
BF = torch.bfloat16
SHAPES = [(20, 184, 320, 32, 128, False, True, "b1.1 fwd 32->128 +stats"), (20, 184, 320, 128, 32, True, False, "b1.1 dgrad 128->32 +res"),
          (20, 92, 160, 48, 192, False, True, "b2.1 fwd 48->192 +stats"), (20, 92, 160, 192, 48, True, False, "b2.1 dgrad 192->48 +res"),
          (20, 368, 640, 16, 32, False, False, "b0.0 dgrad 16->32"), (20, 368, 640, 32, 16, False, True, "b0.0 fwd 32->16 pro +stats"),
          (20, 368, 640, 16, 64, False, True, "b1.0 fwd s2 16->64 pro +stats"), (20, 368, 640, 16, 64, False, True, "b1.0 fwd s2 16->64 plain +stats"),
          (20, 184, 320, 32, 128, False, True, "b2.0 fwd s2 32->128 +stats")]
if os.environ.get("C3_ONLY"):
    SHAPES = [s for s in SHAPES if os.environ["C3_ONLY"] in s[-1]]


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


dbgs = [int(v) for v in sys.argv[1:]] or [0, 1, 2, 4, 8, 3, 6, 7, 15]
for (N, H, W, Cin, Cout, res, stats, tag) in SHAPES:
    stride = 2 if " s2 " in tag else 1
    OH, OW, pt, pl = geo.conv_geometry(H, W, stride)
    x = torch.randn(N * H * W, Cin, device=dev).to(BF); w = torch.randn(Cout * 9 * Cin, device=dev).to(BF)
    y = torch.empty(N * OH * OW, Cout, device=dev, dtype=BF)
    r = torch.randn(N * H * W, Cout, device=dev).to(BF) if res else None
    st = torch.zeros(cabi.MDS_STAT_SLOTS, 2, Cout, device=dev, dtype=torch.float64) if stats else None
    dy, dx, wi = geo.taps_fwd(pt, pl)
    pro = cabi.pro(0)
    if " pro" in tag:
        psc, psh = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev)
        pro = cabi.pro(2, psc, psh)
    a = cabi.make("mds_conv_fwd_args", dtype=1, N=N, IH=H, IW=W, Cin=Cin, OH=OH, OW=OW, Cout=Cout, A=OH, B=OW, oy0=0, ox0=0, os=1, **{"is": stride},
                  ntaps=9, dy=dy, dx=dx, wi=wi, wtaps=9, x=x, w=w, y=y, pro=pro, residual=r, stats=st)
    s = torch.cuda.current_stream().cuda_stream
    flops = 2 * N * OH * OW * 9 * Cin * Cout
    nbytes = (x.numel() + y.numel() * (2 if res else 1)) * 2
    out = []
    lib.fn["dev_set"](cabi.MDS_KNOB_C3, 1)
    t_old = timeit(lambda: lib.call("conv_fwd", a, s))
    lib.fn["dev_set"](cabi.MDS_KNOB_C3, 0)
    for d in dbgs:
        lib.fn["dev_set"](cabi.MDS_KNOB_C3_DBG, d)
        out.append((d, timeit(lambda: lib.call("conv_fwd", a, s))))
    lib.fn["dev_set"](cabi.MDS_KNOB_C3_DBG, 0)
    print(f"{tag:28s} k_conv {t_old:7.1f} us | " + "  ".join(f"dbg{d}: {t:6.1f}" for d, t in out) +
          f" | floor: HBM {nbytes / 5e6:.0f} us at 5 TB/s, MFMA {flops / 2.5e9:.0f} us at peak", flush=True)
