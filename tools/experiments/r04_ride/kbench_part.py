BWG_SHAPES = [  # groups, rows per group, C (wide), K (narrow), tag
    (20, 920, 1152, 192, "s5"), (20, 3680, 672, 112, "s4"), (20, 3680, 384, 96, "s3"), (4, 4600, 576, 192, "3d"),
    (20, 3680, 576, 96, "s4.0 pw"), (20, 920, 672, 192, "s5.0 pwl")]


def bench_bwg():
    """the BatchNorm-backward apply pass with the 1x1 weight gradient riding on it, against the apply pass alone and the
    separate weight-gradient launch it replaces (KB_BWG_BLOCKS: block target)"""
    if os.environ.get("KB_BWG_BLOCKS"):
        lib.check(lib.fn["dev_set"](cabi.MDS_KNOB_BWG_BLOCKS, int(os.environ["KB_BWG_BLOCKS"])), "dev_set")
    for (G, R, C, K, tag) in BWG_SHAPES:
        M = G * R
        u = rnd(M, C); y = rnd(M, C); x = rnd(M, K); dyo = torch.empty_like(y)
        sc = torch.rand(C, device=dev) + 0.5; sh = torch.randn(C, device=dev) * 0.1
        bn = torch.stack([sc, sh, torch.zeros_like(sc), torch.ones_like(sc)]).contiguous()
        lin = torch.rand(3, C, device=dev); coef = torch.rand(3, C, device=dev)
        gate = torch.rand(G, C, device=dev); dpool = torch.randn(G, C, device=dev) * 0.01
        for se in (False, True):
            gs = cabi.gsrc(2 if se else 0, u, gate if se else None, dpool if se else None, rows_per_group=R if se else 0)
            grows = R if se else 0
            slabs = lib.fn["bn_bwd_apply_wg_slabs"](M, C, K, grows, 1 if se else 0, 1)
            part = torch.empty(slabs, C, K, device=dev); dw = torch.zeros(C, K, device=dev)
            a = cabi.make("mds_bn_bwd_apply_wg_args", dtype=1, M=M, C=C, g=gs, y=y, bn=bn, lin=lin, dy=dyo, K=K, x=x,
                          wide_act=1 if se else 0, group_rows=grows, slabs=slabs, part=part)
            f = cabi.make("mds_wg_finish_args", C=C, K=K, slabs=slabs, transpose=1 if se else 0, part=part, dw=dw)
            e = cabi.make("mds_bn_bwd_apply_args", dtype=1, M=M, C=C, g=gs, y=y, bn=bn, coef=coef, dy=dyo)
            nb = 3 * M * C * 2
            kind = "SE " if se else "PLN"
            timeit(f"bwg {kind} {tag} {M}x{C} K={K} slabs={slabs}", lambda: lib.call("bn_bwd_apply_wg", a, stream()), nb, 2 * M * C * K)
            timeit(f"    apply alone", lambda: lib.call("bn_bwd_apply", e, stream()), nb, 0)
            timeit(f"    wg_finish", lambda: lib.call("wg_finish", f, stream()), slabs * C * K * 4, 0)
            if se:
                w = cabi.make("mds_pw_wgrad_args", dtype=1, M=M, K=C, N=K, x=y, dy=x, dw=dw, pro=cabi.pro(4, None, None, gate, R))
            else:
                w = cabi.make("mds_pw_wgrad_args", dtype=1, M=M, K=K, N=C, x=x, dy=dyo, dw=dw, pro=cabi.pro(0))
            timeit(f"    pw_wgrad it replaces", lambda: lib.call("pw_wgrad", w, stream()), (M * K + M * C) * 2, 2 * M * C * K)


def bench_pwd():
    """mds_pw_dgrad (apply pass folded into the expansion's data gradient) against the apply + pw_fwd pair it replaces"""
    if os.environ.get("KB_DBG"):
        lib.check(lib.fn["dev_set"](cabi.MDS_KNOB_WG_DBG, int(os.environ["KB_DBG"])), "dev_set")
    shapes = [(18400, 1152, 192, "s5"), (73600, 672, 112, "s4"), (73600, 576, 96, "s4.0"), (73600, 384, 96, "s3"), (18400, 576, 192, "3d"),
              (294400, 192, 48, "s3.0"), (18400, 672, 112, "s5.0")]
    if os.environ.get("KB_DBG"):
        shapes = shapes[:2]
    for (M, K, N, tag) in shapes:
        u = rnd(M, K); y = rnd(M, K); w = rnd(N, K); res = rnd(M, N); out = torch.empty(M, N, device=dev, dtype=BF); dyo = torch.empty_like(y)
        lin = torch.rand(3, K, device=dev); coef = torch.rand(3, K, device=dev); bn = torch.rand(4, K, device=dev)
        py = rnd(M, N); pbn = torch.rand(4, N, device=dev); st = torch.zeros(SLOTS, 2, N, device=dev, dtype=torch.float64)
        post = cabi.poststat(1, py, pbn, st)
        a = cabi.make("mds_pw_dgrad_args", dtype=1, M=M, K=K, N=N, dyp=cabi.dyp(cabi.gsrc(0, u), y, bn, lin), dy_out=dyo, w=w, y=out, residual=res, post=post)
        nb = (3 * M * K + 3 * M * N) * 2
        timeit(f"pw_dgrad {tag} {M}x{K}->{N} (+res +post +dy)", lambda: lib.call("pw_dgrad", a, stream()), nb, 2 * M * K * N)
        e = cabi.make("mds_bn_bwd_apply_args", dtype=1, M=M, C=K, g=cabi.gsrc(0, u), y=y, bn=bn, coef=coef, dy=dyo)
        timeit(f"    bn_bwd_apply", lambda: lib.call("bn_bwd_apply", e, stream()), 3 * M * K * 2, 0)
        f = cabi.make("mds_pw_fwd_args", dtype=1, M=M, K=K, N=N, x=dyo, w=w, y=out, pro=cabi.pro(0), residual=res, stats=None, post=post)
        timeit(f"    pw_fwd (data gradient, +res +post)", lambda: lib.call("pw_fwd", f, stream()), (M * K + 3 * M * N) * 2, 2 * M * K * N)


