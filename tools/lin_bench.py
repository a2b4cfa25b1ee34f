"""Developer tool: BatchNorm backward of a 1x1 expansion, materialised-dy form (apply + data gradient + weight gradient) against
the linear form (prep + two-pair data gradient; weight gradient with row scale + Gram + fix), each launch timed alone."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ball-action-spotting_amd")]
import torch
from mds import cabi
dev = torch.device("cuda:0"); lib = cabi.load(); BF = torch.bfloat16; SL = cabi.MDS_STAT_SLOTS
def t_us(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
s = lambda: torch.cuda.current_stream().cuda_stream
for (M, Cin, Cmid, tag) in [(18400, 192, 1152, "b5"), (73600, 112, 672, "b4"), (18400, 192, 576, "3d"), (73600, 96, 384, "b3"), (294400, 48, 192, "b3.0"), (73600, 96, 576, "b4.0")]:
    x = torch.randn(M, Cin, device=dev).to(BF); g = torch.randn(M, Cmid, device=dev).to(BF); y = torch.randn(M, Cmid, device=dev).to(BF)
    dy = torch.empty_like(g); dx = torch.empty(M, Cin, device=dev, dtype=BF); W = torch.randn(Cmid, Cin, device=dev)
    bn = torch.rand(4, Cmid, device=dev) + 0.5; coef = torch.rand(3, Cmid, device=dev); lin = torch.rand(3, Cmid, device=dev)
    wt = torch.randn(Cin, Cmid, device=dev).to(BF)
    Kp, K1p = (Cmid + 63) // 64 * 64, (Cin + 63) // 64 * 64
    wcat = torch.empty(Cin, Kp + K1p, device=dev, dtype=BF); bias = torch.empty(Cin, device=dev)
    dW = torch.zeros(Cmid, Cin, device=dev); gram = torch.zeros(Cin, Cin, device=dev); cs = torch.zeros(SL, 2, Cin, device=dev, dtype=torch.float64)
    a_apply = cabi.make("mds_bn_bwd_apply_args", dtype=1, M=M, C=Cmid, g=cabi.gsrc(0, g), y=y, bn=bn, coef=coef, dy=dy)
    a_dg = cabi.make("mds_pw_fwd_args", dtype=1, M=M, K=Cmid, N=Cin, x=dy, w=wt, y=dx, pro=cabi.pro(0), residual=None, stats=None)
    a_prep = cabi.make("mds_bn_lin_prep_args", dtype=1, Cmid=Cmid, Cin=Cin, w=W, lin=lin, wcat=wcat, bias=bias)
    a_dg2 = cabi.make("mds_pw_fwd_args", dtype=1, M=M, K=Cmid, N=Cin, x=g, w=wcat, y=dx, pro=cabi.pro(0), residual=None, stats=None, x1=x, K1=Cin, bias=bias)
    a_wg = cabi.make("mds_pw_wgrad_args", dtype=1, M=M, K=Cin, N=Cmid, x=x, dy=dy, dw=dW, pro=cabi.pro(0))
    a_wg2 = cabi.make("mds_pw_wgrad_args", dtype=1, M=M, K=Cin, N=Cmid, x=x, dy=g, dw=dW, pro=cabi.pro(0), nscale=lin[0].contiguous())
    a_gram = cabi.make("mds_pw_wgrad_args", dtype=1, M=M, K=Cin, N=Cin, x=x, dy=x, dw=gram, pro=cabi.pro(0))
    a_fix = cabi.make("mds_bn_lin_wgrad_args", Cmid=Cmid, Cin=Cin, w=W, lin=lin, gram=gram, colsum=cs, dw=dW)
    a_cs = cabi.make("mds_bn_bwd_reduce_args", dtype=1, M=M, C=Cin, g=cabi.gsrc(0, x), y=x, bn=torch.ones(4, Cin, device=dev), stats=cs)
    T = {k: t_us(lambda a=a, op=op: lib.call(op, a, s())) for k, (op, a) in dict(apply=("bn_bwd_apply", a_apply), dgrad=("pw_fwd", a_dg), prep=("bn_lin_prep", a_prep),
         dgrad2=("pw_fwd", a_dg2), wgrad=("pw_wgrad", a_wg), wgrad_ns=("pw_wgrad", a_wg2), gram=("pw_wgrad", a_gram), fix=("bn_lin_wgrad", a_fix), colsum=("bn_bwd_reduce", a_cs)).items()}
    print(f"{tag:5s} M={M:6d} {Cin:4d}->{Cmid:4d} | main: apply {T['apply']:6.1f} + dgrad {T['dgrad']:6.1f} = {T['apply'] + T['dgrad']:6.1f}  vs  prep {T['prep']:5.1f} + dgrad2 {T['dgrad2']:6.1f} = {T['prep'] + T['dgrad2']:6.1f}"
          f" | side: wgrad {T['wgrad']:6.1f}  vs  wgrad_ns {T['wgrad_ns']:6.1f} + gram {T['gram']:5.1f} + colsum {T['colsum']:5.1f} + fix {T['fix']:5.1f}", flush=True)
