"""Micro-benchmark single kernels of libmds_hip.so at the real layer shapes (developer tool).

  python tools/kbench.py dw_fwd dw_bwd pw_fwd ...        # HIP-event timing, 20 reps
Used under rocprofv3 (--kernel-trace --stats / --pmc ...) to study one kernel in isolation.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ball-action-spotting_amd")]
import torch
from mds import cabi, geometry as geo

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

if os.environ.get("C3_LIB"):      # an experiment build (make c3abl ABL=n)
    cabi.HIP_LIB = os.path.join(ROOT, "ball-action-spotting_amd", "csrc", os.environ["C3_LIB"])
lib = cabi.load()
BF = torch.bfloat16
SLOTS = cabi.MDS_STAT_SLOTS


def timeit(name, fn, nbytes, flops, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print(f"{name:42s} {us:9.1f} us  {nbytes / us / 1e3:8.1f} GB/s  {flops / us / 1e6:8.1f} TFLOP/s", flush=True)


def stream():
    return torch.cuda.current_stream().cuda_stream


def rnd(*shape, dtype=BF):
    return torch.randn(*shape, device=dev).to(dtype)


def bench_dw(which):
    for (N, T, H, W, C, s, kt, tag) in [(20, 1, 46, 80, 384, 1, 1, "s3 46x80x384"), (20, 1, 46, 80, 576, 1, 1, "s4.0 46x80x576"), (20, 1, 46, 80, 672, 1, 1, "s4 46x80x672"), (20, 1, 23, 40, 1152, 1, 1, "s5 23x40x1152"),
                                        (20, 1, 92, 160, 192, 2, 1, "s3.0 92x160x192 s2"), (4, 5, 23, 40, 576, 1, 3, "3d 5x23x40x576"), (4, 11, 23, 40, 576, 1, 3, "3d 11x23x40x576 (config 4)")]:
        OH, OW, pt, pl = geo.conv_geometry(H, W, s)
        x = rnd(N * T * H * W, C); w = torch.randn(C, kt * 9, device=dev) * 0.3
        sc = torch.rand(C, device=dev) + 0.5; sh = torch.randn(C, device=dev) * 0.1
        mean = torch.randn(C, device=dev) * 0.1; rstd = torch.rand(C, device=dev) + 0.5
        y = torch.empty(N * T * OH * OW, C, device=dev, dtype=BF)
        st = torch.zeros(SLOTS, 2, C, device=dev, dtype=torch.float64)
        pro = cabi.pro(int(os.environ.get("KB_MODE", "2")), sc, sh)
        nin, nout = x.numel(), y.numel()
        if which == "dw_fwd":
            a = cabi.make("mds_dw_fwd_args", dtype=1, N=N, T=T, IH=H, IW=W, C=C, OH=OH, OW=OW, stride=s, pad_t=pt, pad_l=pl,
                          kt=kt, x=x, w=w, y=y, pro=pro, stats=None if os.environ.get("KB_NOSTATS") else st)
            timeit(f"dw_fwd {tag}", lambda: lib.call("dw_fwd", a, stream()), (nin + nout) * 2, 2 * 9 * kt * nout)
        else:
            dy = rnd(N * T * OH * OW, C); g = torch.empty_like(x); dw = torch.zeros(C, kt * 9, device=dev)
            a = cabi.make("mds_dw_bwd_args", dtype=1, N=N, T=T, IH=H, IW=W, C=C, OH=OH, OW=OW, stride=s, pad_t=pt, pad_l=pl,
                          kt=kt, x=x, dy=dy, w=w, g=g, dw=dw, pro=pro, mean=mean, rstd=rstd, stats=st)
            timeit(f"dw_bwd {tag}", lambda: lib.call("dw_bwd", a, stream()), (2 * nin + nout) * 2, 4 * 9 * kt * nout)


PW_SHAPES = [  # M, K, N, pro, tag
    (20 * 184 * 320, 128, 32, 2, "b1.1 pwl 128->32"), (20 * 92 * 160, 48, 192, 0, "b3.0 pw 48->192"),
    (20 * 46 * 80, 96, 384, 0, "b3.x pw 96->384"), (20 * 46 * 80, 112, 672, 0, "b4.x pw 112->672"), (20 * 46 * 80, 672, 112, 3, "b4.x pwl 672->112"),
    (20 * 23 * 40, 192, 1152, 0, "b5.x pw 192->1152"), (20 * 23 * 40, 1152, 192, 3, "b5.x pwl 1152->192"),
    (4 * 5 * 23 * 40, 192, 576, 0, "3d pw 192->576"), (4 * 5 * 23 * 40, 576, 192, 3, "3d pwl 576->192"),
]


def bench_pw(which):
    for (M, K, N, mode, tag) in PW_SHAPES:
        x = rnd(M, K); w = rnd(N, K); y = torch.empty(M, N, device=dev, dtype=BF)
        sc = torch.rand(K, device=dev) + 0.5; sh = torch.randn(K, device=dev) * 0.1
        rpg = M // 20
        gate = torch.rand(20, K, device=dev)
        st = torch.zeros(SLOTS, 2, N, device=dev, dtype=torch.float64)
        pro = cabi.pro(mode, sc, sh, gate, rpg)
        if which == "pw_fwd":
            a = cabi.make("mds_pw_fwd_args", dtype=1, M=M, K=K, N=N, x=x, w=w, y=y, pro=pro, residual=None,
                          stats=None if os.environ.get("KB_NOSTATS") else st)
            timeit(f"pw_fwd {tag}", lambda: lib.call("pw_fwd", a, stream()), (M * K + M * N + N * K) * 2, 2 * M * K * N)
        else:
            dy = rnd(M, N); dw = torch.zeros(N, K, device=dev)
            a = cabi.make("mds_pw_wgrad_args", dtype=1, M=M, K=K, N=N, x=x, dy=dy, dw=dw, pro=pro)
            timeit(f"pw_wgrad {tag}", lambda: lib.call("pw_wgrad", a, stream()), (M * K + M * N) * 2, 2 * M * K * N)


KH_SHAPES = [  # M, K, N, pro, tag: the K-streaming class (projections forward = data gradients of the expansions)
    (20 * 23 * 40, 1152, 192, 3, "b5.x pwl 1152->192"), (20 * 23 * 40, 672, 192, 3, "b5.0 pwl 672->192"),
    (20 * 46 * 80, 672, 112, 3, "b4.x pwl 672->112"), (20 * 46 * 80, 576, 112, 3, "b4.0 pwl 576->112"),
    (20 * 46 * 80, 384, 96, 3, "b3.x pwl 384->96"), (20 * 46 * 80, 192, 96, 3, "b3.0 pwl 192->96"),
    (4 * 5 * 23 * 40, 576, 192, 3, "3d pwl 576->192"), (20 * 23 * 40, 192, 192, 0, "proj2d 192->192"),
    (44 * 23 * 40, 1152, 192, 3, "config 4: b5.x pwl 1152->192 @40480"), (44 * 23 * 40, 576, 192, 3, "config 4: 3d pwl 576->192 @40480"),
    (44 * 23 * 40, 192, 192, 0, "config 4: proj2d 192->192 @40480"),
]


def bench_pwk():
    """old general kernel (knob 18 = 1) against the K-streaming kernel at several prefetch distances; forward (BN + SiLU +
    gate prologue, statistics) and data-gradient form (residual + MASK post statistics)"""
    for (M, K, N, mode, tag) in KH_SHAPES:
        x = rnd(M, K); w = rnd(N, K); y = torch.empty(M, N, device=dev, dtype=BF)
        sc = torch.rand(K, device=dev) + 0.5; sh = torch.randn(K, device=dev) * 0.1
        rpg = M // 20
        gate = torch.rand(20, K, device=dev)
        st = torch.zeros(SLOTS, 2, N, device=dev, dtype=torch.float64)
        nbytes, flops = (M * K + M * N + N * K) * 2, 2 * M * K * N
        wfr = frag_pack(lib, w)
        fwd = cabi.make("mds_pw_fwd_args", dtype=1, M=M, K=K, N=N, x=x, w=w, y=y, pro=cabi.pro(mode, sc, sh, gate, rpg), residual=None, stats=st, w_frag=wfr)
        ys = rnd(M, N); res = rnd(M, N); bn = torch.rand(4, N, device=dev) + 0.5; mask = torch.ones(20, device=dev)
        dg = cabi.make("mds_pw_fwd_args", dtype=1, M=M, K=K, N=N, x=x, w=w, y=y, pro=cabi.pro(0), residual=res, stats=None,
                       post=cabi.poststat(2, ys, bn, st, mask, rpg), w_frag=wfr)
        for name, a, nb in (("fwd", fwd, nbytes), ("dgrad", dg, nbytes + 2 * M * N * 2)):
            lib.fn["dev_set"](18, 1)
            timeit(f"{name:5s} {tag} general", lambda: lib.call("pw_fwd", a, stream()), nb, flops)
            lib.fn["dev_set"](18, 2)
            timeit(f"{name:5s} {tag} kstream", lambda: lib.call("pw_fwd", a, stream()), nb, flops)
            lib.fn["dev_set"](18, 0)


CONV_SHAPES = [(20, 368, 640, 32, 16, 1, 2, "b0.0 32->16"), (20, 368, 640, 16, 64, 2, 2, "b1.0 16->64 s2"),
               (20, 184, 320, 32, 128, 1, 0, "b1.1 32->128"), (20, 184, 320, 32, 128, 2, 0, "b2.0 32->128 s2"),
               (20, 92, 160, 48, 192, 1, 0, "b2.1 48->192")]


def bench_conv(which):
    for (N, H, W, Cin, Cout, s, mode, tag) in CONV_SHAPES:
        if os.environ.get("KB_MODE"):
            mode = int(os.environ["KB_MODE"])
        OH, OW, pt, pl = geo.conv_geometry(H, W, s)
        dy_, dx_, wi = geo.taps_fwd(pt, pl)
        x = rnd(N * H * W, Cin); w = rnd(Cout * 9 * Cin)
        sc = torch.rand(Cin, device=dev) + 0.5; sh = torch.randn(Cin, device=dev) * 0.1
        pro = cabi.pro(mode, sc, sh)
        flops = 2 * N * OH * OW * 9 * Cin * Cout
        if which == "conv_fwd":
            y = torch.empty(N * OH * OW, Cout, device=dev, dtype=BF); st = torch.zeros(SLOTS, 2, Cout, device=dev, dtype=torch.float64)
            a = cabi.make("mds_conv_fwd_args", dtype=1, N=N, IH=H, IW=W, Cin=Cin, OH=OH, OW=OW, Cout=Cout, A=OH, B=OW, oy0=0,
                          ox0=0, os=1, **{"is": s}, ntaps=9, dy=dy_, dx=dx_, wi=wi, wtaps=9, x=x, w=w, y=y, pro=pro,
                          residual=None, stats=None if os.environ.get("KB_NOSTATS") else st)
            timeit(f"conv_fwd {tag}", lambda: lib.call("conv_fwd", a, stream()), (x.numel() + y.numel()) * 2, flops)
        else:
            dyt = rnd(N * OH * OW, Cout); dw = torch.zeros(Cout, Cin, 3, 3, device=dev)
            a = cabi.make("mds_conv_wgrad_args", dtype=1, N=N, IH=H, IW=W, Cin=Cin, OH=OH, OW=OW, Cout=Cout, **{"is": s},
                          ntaps=9, dy=dy_, dx=dx_, wi=wi, wtaps=9, x=x, dyt=dyt, dw=dw, pro=pro)
            timeit(f"conv_wgrad {tag}", lambda: lib.call("conv_wgrad", a, stream()), (x.numel() + dyt.numel()) * 2, flops)


DGRAD_SHAPES = [(20, 368, 640, 32, 16, 1, False, "b0.0 dgrad 16->32"), (20, 368, 640, 16, 64, 2, False, "b1.0 dgrad 64->16 s2"),
                (20, 184, 320, 32, 128, 1, True, "b1.1 dgrad 128->32 +res"), (20, 184, 320, 32, 128, 2, False, "b2.0 dgrad 128->32 s2"),
                (20, 92, 160, 48, 192, 1, True, "b2.1 dgrad 192->48 +res")]


def bench_conv_dgrad():
    """data gradients exactly as engine._conv_dgrad launches them (forward-conv shapes in the table)"""
    for (N, H, W, Cin, Cout, s, res, tag) in DGRAD_SHAPES:
        OH, OW, pt, pl = geo.conv_geometry(H, W, s)
        dyb = rnd(N * OH * OW, Cout); w = rnd(Cin * 9 * Cout); dxb = torch.empty(N * H * W, Cin, device=dev, dtype=BF)
        r = rnd(N * H * W, Cin) if res else None
        common = dict(dtype=1, N=N, IH=OH, IW=OW, Cin=Cout, OH=H, OW=W, Cout=Cin, wtaps=9, x=dyb, w=w, y=dxb, pro=cabi.pro(0),
                      residual=r, stats=None)
        flops = 2 * N * OH * OW * 9 * Cin * Cout
        nbytes = (dyb.numel() + dxb.numel() * (2 if res else 1)) * 2
        if s == 1:
            dy, dx, wi = geo.taps_dgrad_s1()
            a = cabi.make("mds_conv_fwd_args", A=H, B=W, oy0=0, ox0=0, os=1, **{"is": 1}, ntaps=9, dy=dy, dx=dx, wi=wi, **common)
            timeit(f"conv {tag}", lambda: lib.call("conv_fwd", a, stream()), nbytes, flops)
            continue
        par = []
        for py in range(2):
            for px in range(2):
                dy, dx, wi = geo.taps_dgrad_s2(py, px, pt, pl)
                par.append((py, px, dy, dx, wi, (H - py + 1) // 2, (W - px + 1) // 2))
        one = cabi.make("mds_conv_fwd_args", A=max(p[5] for p in par), B=max(p[6] for p in par), oy0=0, ox0=0, os=2, **{"is": 1},
                        ntaps=sum(len(p[2]) for p in par), dy=sum((p[2] for p in par), []), dx=sum((p[3] for p in par), []),
                        wi=sum((p[4] for p in par), []), ngroups=4, g_ntaps=[len(p[2]) for p in par],
                        g_oy0=[p[0] for p in par], g_ox0=[p[1] for p in par], g_A=[p[5] for p in par], g_B=[p[6] for p in par], **common)
        four = [cabi.make("mds_conv_fwd_args", A=A, B=B_, oy0=py, ox0=px, os=2, **{"is": 1}, ntaps=len(dy), dy=dy, dx=dx, wi=wi, **common)
                for (py, px, dy, dx, wi, A, B_) in par]
        timeit(f"conv {tag} (tap groups)", lambda: lib.call("conv_fwd", one, stream()), nbytes, flops)
        timeit(f"conv {tag} (4 launches)", lambda: [lib.call("conv_fwd", a4, stream()) for a4 in four], nbytes, flops)


def bench_pw_as_conv():
    """the 1x1 expansion GEMMs through the persistent convolution kernel (one tap): slab-resident weights, streamed rows"""
    for (M, K, N, mode, tag) in PW_SHAPES:
        if mode not in (0, 2) or M % 16:
            continue
        x = rnd(M, K); w = rnd(N, K); y = torch.empty(M, N, device=dev, dtype=BF)
        st = torch.zeros(SLOTS, 2, N, device=dev, dtype=torch.float64)
        sc = torch.rand(K, device=dev) + 0.5; sh = torch.randn(K, device=dev) * 0.1
        a = cabi.make("mds_conv_fwd_args", dtype=1, N=1, IH=M // 16, IW=16, Cin=K, OH=M // 16, OW=16, Cout=N, A=M // 16, B=16, oy0=0,
                      ox0=0, os=1, **{"is": 1}, ntaps=1, dy=[0], dx=[0], wi=[0], wtaps=1, x=x, w=w, y=y, pro=cabi.pro(mode, sc, sh),
                      residual=None, stats=st)
        timeit(f"pw via conv_q {tag}", lambda: lib.call("conv_fwd", a, stream()), (M * K + M * N + N * K) * 2, 2 * M * K * N)
        a2 = cabi.make("mds_pw_fwd_args", dtype=1, M=M, K=K, N=N, x=x, w=w, y=y, pro=cabi.pro(mode, sc, sh), residual=None, stats=st)
        timeit(f"pw_fwd       {tag}", lambda: lib.call("pw_fwd", a2, stream()), (M * K + M * N + N * K) * 2, 2 * M * K * N)


def bench_se():
    for (G, R, C) in [(20, 3680, 672), (20, 920, 1152), (20, 3680, 384), (4, 4600, 576)]:
        bench_se1(G, R, C)


def bench_se1(G, R, C):
    M = G * R
    u = rnd(M, C); y = rnd(M, C)
    sc = torch.rand(C, device=dev) + 0.5; sh = torch.randn(C, device=dev) * 0.1
    mean = torch.randn(C, device=dev) * 0.1; rstd = torch.rand(C, device=dev) + 0.5
    dgate = torch.zeros(G, C, device=dev, dtype=torch.float64); bns = torch.zeros(G, 64, 4, C, device=dev)
    a = cabi.make("mds_se_bwd_reduce_args", dtype=1, groups=G, rows_per_group=R, C=C, u=u, y=y, scale=sc, shift=sh,
                  dgate=dgate, mean=mean, rstd=rstd, bnsums=bns)
    timeit(f"se_bwd_reduce fused-bn {G}x{R}x{C}", lambda: lib.call("se_bwd_reduce", a, stream()), 2 * M * C * 2, 0)
    b = cabi.make("mds_se_bwd_reduce_args", dtype=1, groups=G, rows_per_group=R, C=C, u=u, y=y, scale=sc, shift=sh,
                  dgate=dgate)
    timeit(f"se_bwd_reduce plain     {G}x{R}x{C}", lambda: lib.call("se_bwd_reduce", b, stream()), 2 * M * C * 2, 0)
    pooled = torch.zeros(G, C, device=dev, dtype=torch.float64); act = torch.empty_like(y)
    c = cabi.make("mds_se_pool_args", dtype=1, groups=G, rows_per_group=R, C=C, y=y, scale=sc, shift=sh, pooled=pooled, act=act)
    timeit(f"se_pool (+act)          {G}x{R}x{C}", lambda: lib.call("se_pool", c, stream()), 2 * M * C * 2, 0)
    st = torch.zeros(SLOTS, 2, C, device=dev, dtype=torch.float64); bn = torch.stack([sc, sh, mean, rstd]).contiguous()
    d = cabi.make("mds_bn_bwd_reduce_args", dtype=1, M=M, C=C, g=cabi.gsrc(1, u), y=y, bn=bn, stats=st)
    timeit(f"bn_bwd_reduce silu      {G}x{R}x{C}", lambda: lib.call("bn_bwd_reduce", d, stream()), 2 * M * C * 2, 0)
    coef = torch.rand(3, C, device=dev); dyo = torch.empty_like(y)
    e = cabi.make("mds_bn_bwd_apply_args", dtype=1, M=M, C=C, g=cabi.gsrc(1, u), y=y, bn=bn, coef=coef, dy=dyo)
    timeit(f"bn_bwd_apply silu       {G}x{R}x{C}", lambda: lib.call("bn_bwd_apply", e, stream()), 3 * M * C * 2, 0)


def bench_stem():
    N, H, W = 20, 736, 1280
    OH, OW, pt, pl = geo.conv_geometry(H, W, 2)
    x = torch.rand(N, 3, H, W, device=dev); dy = rnd(N * OH * OW, 32); dw = torch.zeros(32, 27, device=dev)
    a = cabi.make("mds_stem_wgrad_args", dtype=1, N=N, H=H, W=W, OH=OH, OW=OW, Cout=32, pad_t=pt, pad_l=pl, x=x, dy=dy, dw=dw)
    timeit("stem_wgrad 20x3x736x1280", lambda: lib.call("stem_wgrad", a, stream()), x.numel() * 4 + dy.numel() * 2, 2 * N * OH * OW * 27 * 32)


def bench_copy():
    n = 256 * 1024 * 1024
    a = torch.empty(n, device=dev, dtype=torch.uint8); b = torch.empty_like(a)
    timeit("torch copy 256MB (HBM reference)", lambda: b.copy_(a), 2 * n, 0)


if __name__ == "__main__":
    for knob, env in ((0, "KB_CONV_BLOCKS"), (1, "KB_DW_ORDER"), (2, "KB_PW_WRES")):
        if os.environ.get(env):
            lib.check(lib.fn["dev_set"](knob, int(os.environ[env])), "dev_set")
    todo = sys.argv[1:] or ["copy", "dw_fwd", "dw_bwd", "pw_fwd", "pw_wgrad", "conv_fwd", "conv_wgrad"]
    for t in todo:
        if t == "copy":
            bench_copy()
        elif t == "stem":
            bench_stem()
        elif t == "se":
            bench_se()
        elif t.startswith("dw"):
            bench_dw(t)
        elif t == "pwk":
            bench_pwk()
        elif t == "pw_as_conv":
            bench_pw_as_conv()
        elif t.startswith("pw"):
            bench_pw(t)
        elif t == "conv_dgrad":
            bench_conv_dgrad()
        else:
            bench_conv(t)
