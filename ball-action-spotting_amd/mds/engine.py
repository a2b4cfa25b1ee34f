"""Launch planner / executor for the MultiDimStacker hot path on MI355X.

For one (shape, dtype, mode) the planner walks the module once and records the complete forward
and backward *launch schedules* as pre-filled C-ABI argument structs (include/mds.h) over
pre-allocated device buffers — no tracing compiler, no per-step allocation.  A training step is
then: one memset of the statistics arena, one weight-pack launch, ~300 forward launches, and
(later) ~500 backward launches, all asynchronous on the caller's HIP stream.

Data layout in HBM (all channels-last "rows", C contiguous):
  frames      fp32  (B*S, 3, H, W) planes  — the reference's x.view(b*S, 3, h, w), read in place
  activations T     [N*H*W][C] (2D) / [B*S*h*w][C] (3D), T = bf16 (autocast) or fp32
  every conv keeps its *raw* output y plus fp32 per-channel sums; BatchNorm(+SiLU)(+SE gate) is
  applied by the consumer while loading ("prologue"), so each tensor is written once and the
  normalised copy never exists in memory.  Block outputs (after BN3 + DropPath + skip) are
  materialised because two consumers need them.
Reference mapping: forward_2d / forward_3d / forward_head of
/root/reference/src/models/multidim_stacker.py:210-243 and timm EfficientNetFeatures.forward.
"""
from __future__ import annotations

import contextlib
import os
import ctypes as C
from typing import Dict, List, Optional

import torch

from . import cabi, geometry as geo
from .cabi import MDS_STAT_SLOTS as SLOTS

PRO_NONE, PRO_AFFINE, PRO_BN_SILU, PRO_BN_GATE, PRO_GATE = 0, 1, 2, 3, 4
G_PLAIN, G_SILU, G_SE, G_MASK = 0, 1, 2, 3
POST_NONE, POST_PLAIN, POST_MASK, POST_SILU = 0, 1, 2, 3
EPI_NONE, EPI_AFFINE, EPI_BN_SILU = 0, 1, 2


class Grad:
    """A gradient tensor handed from one backward closure to the next.  `reduced` names the BatchNorm layer
    whose backward sums (sum g, sum g*xhat) the PRODUCING kernel already took in its epilogue (mds_poststat_t)."""
    __slots__ = ("buf", "reduced")

    def __init__(self, buf, reduced=None):
        self.buf, self.reduced = buf, reduced


class Lazy:
    """A buffer request that is bound to device memory when the plan is finalised."""
    __slots__ = ("kind", "numel", "dtype", "tensor", "parent", "off")

    def __init__(self, kind, numel, dtype, parent=None, off=0):
        self.kind, self.numel, self.dtype = kind, int(numel), dtype
        self.tensor, self.parent, self.off = None, parent, int(off)

    def sub(self, off, n):
        return Lazy("sub", n, self.dtype, self, off)

    def resolve(self):
        if self.tensor is None:
            assert self.kind == "sub", f"unbound buffer of kind {self.kind}"
            self.tensor = self.parent.resolve()[self.off:self.off + self.numel]
        return self.tensor


class P:
    """Reference to a live parameter / buffer tensor of the module."""
    __slots__ = ("t",)

    def __init__(self, t):
        self.t = t

    def resolve(self):
        return self.t.detach()


class BNL:
    """One BatchNorm layer: parameters + the fp32 side buffers the kernels exchange."""

    def __init__(self, pb, mod, C_, count):
        self.mod, self.C, self.count = mod, C_, int(count)
        self.buf = pb.f32(4 * C_)                      # scale | shift | mean | rstd
        self.stats = pb.zero_fwd64(SLOTS * 2 * C_) if pb.batch_stats else None      # fp64 slots (include/mds.h)
        self.bstats = pb.zero_bwd64(SLOTS * 2 * C_) if pb.need_grad else None      # fp64 slots (include/mds.h)
        self.coef = pb.f32(3 * C_) if pb.need_grad else None
        self.lin = pb.f32(3 * C_) if pb.need_grad else None     # A, B, D of dy = A*g + B*y + D (mds_dyp_t)
        # fp32 plans: sum_g/M, sum_gx/M and the batch mean in fp64 for the apply pass (mds_bn_bwd_finalize_args.coef64)
        self.coef64 = pb._own(3 * C_, torch.float64) if (pb.need_grad and pb.code == cabi.MDS_F32) else None

    scale = property(lambda s: s.buf.sub(0, s.C))
    shift = property(lambda s: s.buf.sub(s.C, s.C))
    mean = property(lambda s: s.buf.sub(2 * s.C, s.C))
    rstd = property(lambda s: s.buf.sub(3 * s.C, s.C))

    def pro(self, mode=PRO_BN_SILU, gate=None, rpg=0):
        return dict(mode=mode, scale=self.scale, shift=self.shift, gate=gate, rows_per_group=rpg)

    def finalize(self, pb, seg):
        m = self.mod
        if not pb.batch_stats:      # eval: running statistics only — every layer of the plan in one table-driven launch
            if not any(e is self for e in pb.eval_bn):
                pb.eval_bn.append(self)
            return
        pb._bn_buffers += [m.running_mean, m.running_var, m.num_batches_tracked]
        pb.op(seg, "bn_finalize", C=self.C, count=self.count, stats=self.stats, gamma=P(m.weight), beta=P(m.bias),
              eps=float(m.eps), momentum=float(m.momentum if m.momentum is not None else 0.1),
              training=int(pb.batch_stats), running_mean=P(m.running_mean), running_var=P(m.running_var),
              num_batches_tracked=P(m.num_batches_tracked) if pb.batch_stats else None, out=self.buf)

    def bwd_reduce(self, pb, seg, gsrc, y):
        pb.op(seg, "bn_bwd_reduce", dtype=pb.code, M=self.count, C=self.C, g=gsrc, y=y, bn=self.buf, stats=self.bstats)

    def bwd_finalize(self, pb, seg, frozen=False):
        pb.op(seg, "bn_bwd_finalize", C=self.C, count=self.count, stats=self.bstats, gamma=P(self.mod.weight), bn=self.buf,
              dgamma=None if frozen else pb.grad(self.mod.weight), dbeta=None if frozen else pb.grad(self.mod.bias),
              coef=self.coef, lin=self.lin, batch_stats=int(pb.batch_stats), fwd_stats=self.stats if self.coef64 is not None else None,
              coef64=self.coef64)

    def backward(self, pb, seg, gsrc, y, dy, reduce=True, frozen=False):
        """emit reduce / finalize / apply (dy materialised); `gsrc` describes how g is derived (mds_gsrc_t)."""
        if reduce:
            self.bwd_reduce(pb, seg, gsrc, y)
        self.bwd_finalize(pb, seg, frozen)
        pb.op(seg, "bn_bwd_apply", dtype=pb.code, M=self.count, C=self.C, g=gsrc, y=y, bn=self.buf, coef=self.coef, dy=dy,
              coef64=self.coef64)

    def head(self, y, mode, mask=None, rpg=0):
        """what a producing data-gradient GEMM needs to take this layer's backward sums in its epilogue"""
        return dict(bn=self, y=y, mode=mode, mask=mask, rpg=rpg)

    def post(self, head):
        return dict(_struct="mds_poststat_t", mode=head["mode"], y=head["y"], bn=self.buf, mask=head["mask"],
                    rows_per_group=head["rpg"], stats=self.bstats)


def op_cost(name, kw, es):
    """(algorithmic HBM bytes, flops) of one launch: every operand tensor read or written once
    (layer-granular compulsory traffic).  es = activation element size."""
    g = kw.get
    if name == "pw_fwd":
        M, K, N = g("M"), g("K"), g("N")
        return (M * K + M * N * (2 if g("residual") is not None else 1) + N * K) * es, 2 * M * K * N
    if name == "pw_wgrad":
        M, K, N = g("M"), g("K"), g("N")
        return (M * K + M * N) * es + N * K * 4, 2 * M * K * N
    if name == "conv_fwd" and g("ngroups"):
        pts = [a_ * b_ for a_, b_ in zip(g("g_A"), g("g_B"))]
        return (g("N") * g("IH") * g("IW") * g("Cin") + g("N") * sum(pts) * g("Cout")) * es + g("Cout") * g("wtaps") * g("Cin") * es, \
            2 * g("N") * sum(p_ * t_ for p_, t_ in zip(pts, g("g_ntaps"))) * g("Cin") * g("Cout")
    if name == "conv_fwd":
        frac = 1.0 / (g("os") * g("os"))
        nin = g("N") * g("IH") * g("IW") * g("Cin") * frac
        nout = g("N") * g("A") * g("B") * g("Cout")
        return (nin + nout * (2 if g("residual") is not None else 1)) * es + g("Cout") * g("wtaps") * g("Cin") * es, \
            2 * g("N") * g("A") * g("B") * g("ntaps") * g("Cin") * g("Cout")
    if name == "conv_wgrad":
        return (g("N") * g("IH") * g("IW") * g("Cin") + g("N") * g("OH") * g("OW") * g("Cout")) * es + g("Cout") * 9 * g("Cin") * 4, \
            2 * g("N") * g("OH") * g("OW") * 9 * g("Cin") * g("Cout")
    if name == "stem_fwd":
        px = g("N") * g("OH") * g("OW")
        return g("N") * 3 * g("H") * g("W") * 4 + px * g("Cout") * es, 2 * 27 * g("Cout") * px
    if name == "stem_wgrad":
        px = g("N") * g("OH") * g("OW")
        nwide = 2 if g("dyp") else 1      # dy formed on load (mds_dyp_t): the gradient source u AND the raw stem output y are read
        return g("N") * 3 * g("H") * g("W") * 4 + nwide * px * g("Cout") * es, 2 * 27 * g("Cout") * px
    if name == "dw_fwd":
        nin = g("N") * g("T") * g("IH") * g("IW") * g("C")
        nout = g("N") * g("T") * g("OH") * g("OW") * g("C")
        return (nin + nout) * es, 2 * 9 * g("kt") * nout
    if name == "dw_bwd":
        nin = g("N") * g("T") * g("IH") * g("IW") * g("C")
        nout = g("N") * g("T") * g("OH") * g("OW") * g("C")
        return (2 * nin + nout) * es, 4 * 9 * g("kt") * nout
    if name == "bn_res":
        return g("M") * g("C") * es * (3 if g("shortcut") is not None else 2), 4 * g("M") * g("C")
    if name in ("se_pool", "gem_fwd"):
        n = g("groups") * g("rows_per_group") * g("C")
        return n * es * (2 if g("act") is not None else 1), 8 * n
    if name in ("se_bwd_reduce", "bn_bwd_reduce"):
        n = g("M") * g("C") if name == "bn_bwd_reduce" else g("groups") * g("rows_per_group") * g("C")
        return 2 * n * es, 10 * n
    if name in ("bn_bwd_apply", "gem_bwd"):
        n = g("M") * g("C") if name == "bn_bwd_apply" else g("groups") * g("rows_per_group") * g("C")
        return 3 * n * es, 12 * n
    return 0, 0


def gsrc(mode, u, gate=None, dpooled=None, mask=None, rpg=0):
    return dict(_struct="mds_gsrc_t", mode=mode, u=u, gate=gate, dpooled=dpooled, mask=mask, rows_per_group=rpg)


def _hip_path():
    with open("/proc/self/maps") as f:
        for line in f:
            if "libamdhip64" in line:
                return line.split()[-1]
    raise RuntimeError("libamdhip64 is not mapped in this process")


class Plan:
    """The recorded schedules + buffers for one configuration."""

    def __init__(self, module, lib, device, kind, B, T, H, W, code, training, need_grad, enc_grad, ingest=None):
        self.lib, self.device, self.kind = lib, device, kind
        self.ingest = ingest        # 2D plans: (src_h, src_w, nsrc) - the encoder reads raw uint8 frames (pad + /255 + TTA flip fused in the stem);
                                    # tail plans: ("probs", tta) - the head also writes the TTA-mean of the sigmoids (mds.predict)
        self.eval_bn = []           # the BNL of every BatchNorm of an eval-mode plan
        self.code = code
        self.tdt = torch.bfloat16 if code == cabi.MDS_BF16 else torch.float32
        self.training = training
        self.batch_stats = training            # F.batch_norm(training=True) semantics
        self.update_running = training
        self.need_grad = need_grad
        self.enc_grad = enc_grad and need_grad
        self.m = module
        # MDS_FUSE_BN_BWD=1 (default): the data-gradient GEMM that produces a block's input gradient also takes the sums of the
        # BatchNorm backward that consumes it (mds_poststat_t): 22 bn_bwd_reduce launches less.  0: every reduce is its own launch.
        # (Forming dy on load inside the GEMMs - all layers / narrow layers only - and the linear form of BatchNorm backward were
        # measured slower in rounds 2 and 3 and are no longer part of the product: DESIGN 5.)
        # fp32 plans (speed is not their point, the 1e-3 parity bar is): the sums of the residual-stream BatchNorms - sum g cancels
        # to ~1e-3 of its absolute mass over 1.2 - 5.9 M rows - always go through mds_bn_bwd_reduce, whose fp32 instantiation
        # accumulates per thread and per block in fp64 (fp32 block partials put the worst batch-4 bias at 1.2 - 1.5e-3).
        self.fuse_bn_bwd = os.environ.get("MDS_FUSE_BN_BWD", "1") != "0" and code == cabi.MDS_BF16
        # inference plans (eval-mode BatchNorm, no gradient): producers store activated outputs (mds_epi_t)
        self.eval_epilogues = (not training) and (not need_grad) and os.environ.get("MDS_EVAL_EPI", "1") == "1"
        self.in_flight = False
        self.generation = 0      # bumped by every grad-enabled forward: a stale autograd node must not run
        self.profile = None      # list -> run() brackets every launch with HIP events
        self._lazy: List[Lazy] = []
        self._zf, self._zb = 0, 0
        self.segs: Dict[str, list] = {"pack": [], "f2d": [], "f3d": [], "fhead": [], "bhead": [], "b3d": [], "b2d": []}
        self._recs: Dict[str, list] = {"2d": [], "3d": [], "head": []}
        self.pack_jobs = []
        self._se_pending = []       # squeeze-excite parameter-gradient launches deferred to the end of their gradient bucket
        self.se_table = os.environ.get("MDS_SE_PARAMS_TABLE", "1") == "1"
        self._bn_buffers = []    # running_mean / running_var / num_batches_tracked of every BatchNorm this plan updates
        self.taps = []           # block outputs in forward order: dict(tag, buf, bn (raw tensor read through BN+SiLU) or None, rows, C)
        self.masks = []          # (Lazy view, keep_prob)
        self._mask_total = 0
        self.mask_arena = Lazy("own", 0, torch.float32)
        self.zf_arena = Lazy("own", 0, torch.float32)
        self.zb_arena = Lazy("own", 0, torch.float32)
        self.zb64_arena = Lazy("own", 0, torch.float64)     # backward BatchNorm sums (fp64 slots)
        self.zf64_arena = Lazy("own", 0, torch.float64)     # forward BatchNorm statistics (fp64 slots)
        self._zb64, self._zf64 = 0, 0
        # flat gradient arena over all parameters, in parameter order
        self.params = list(module.parameters())
        self.poff, off = {}, 0
        for p in self.params:
            self.poff[id(p)] = off
            off += p.numel()
        self.grad_arena = Lazy("own", off if need_grad else 0, torch.float32)
        self.B, self.T, self.H, self.W = B, T, H, W
        self._build()
        self._finalize()

    # ------------------------------------------------------------------ buffer requests
    def _own(self, n, dtype):
        l = Lazy("own", n, dtype)
        self._lazy.append(l)
        return l

    def act(self, rows, ch):
        return self._own(int(rows) * ch, self.tdt)

    def f32(self, n):
        return self._own(n, torch.float32)

    def _split(self, M, K, N_):
        """split-K fields of an inference-plan pw_fwd (mds_pw_fwd_args.split): the factor comes from the library's own rule; the
        partial buffer and the (self-resetting) tickets are shared by every layer of the plan - launches of one stream"""
        if not self.eval_epilogues:
            return {}
        S = int(self.lib.fn["pw_fwd_split"](int(M), int(K), int(N_), int(self.code)))
        if S <= 1:
            return {}
        if getattr(self, "_split_part", None) is None:
            self._split_part = self._own(0, torch.float32)
            self._split_ticket = Lazy("own0", 0, torch.int32)
            self._lazy.append(self._split_ticket)
        self._split_part.numel = max(self._split_part.numel, S * int(M) * int(N_))
        self._split_ticket.numel = max(self._split_ticket.numel, -(-int(M) // cabi.MDS_PW_SPLIT_TILE_ROWS) * -(-int(N_) // 128) * cabi.MDS_PW_SPLIT_TICKET_STRIDE)
        return dict(split=S, split_part=self._split_part, split_ticket=self._split_ticket)

    def zero_fwd(self, n):
        l = self.zf_arena.sub(self._zf, n)
        self._zf += n
        return l

    def zero_bwd(self, n):
        l = self.zb_arena.sub(self._zb, n)
        self._zb += n
        return l

    def zero_fwd64(self, n):
        l = self.zf64_arena.sub(self._zf64, n)
        self._zf64 += n
        return l

    def zero_bwd64(self, n):
        l = self.zb64_arena.sub(self._zb64, n)
        self._zb64 += n
        return l

    def grad(self, p):
        return self.grad_arena.sub(self.poff[id(p)], p.numel())

    def mask(self, n, rate):
        if not self.training or not rate:
            return None
        l = self.mask_arena.sub(self._mask_total, n)
        self.masks.append((self._mask_total, n, 1.0 - rate))
        self._mask_total += n
        return l

    def pack(self, param, kind, O, I, taps):
        n = O * 32 if kind == cabi.MDS_PACK_STEM else O * I * taps
        if kind in (cabi.MDS_PACK_FRAG_OI, cabi.MDS_PACK_FRAG_IO):      # MFMA-fragment order of w[N][K], zero padded (include/mds.h)
            N_, K = (O, I) if kind == cabi.MDS_PACK_FRAG_OI else (I, O)
            n = -(-K // 32) * -(-N_ // 16) * 512
        dst = self._own(n, torch.float32 if kind == cabi.MDS_PACK_IO_F32 else self.tdt)
        self.pack_jobs.append((param, dst, kind, O, I, taps))
        return dst

    def op(self, seg, name, **kw):
        self.segs[seg].append((name, kw))

    # ------------------------------------------------------------------ graph construction
    def _build(self):
        m = self.m
        B, T, H, W = self.B, self.T, self.H, self.W
        S = T // m.stack_size
        N = B * S
        if self.kind in ("full", "2d"):
            self.x_in = Lazy("input", B * T * H * W, torch.float32)   # bound to the caller's tensor per call
            feat, h, w = self._build_2d(N, H, W)
            self.feat, self.h, self.w = feat, h, w
        else:
            h, w = H, W  # for '3d' / 'head' plans H, W are the feature-map extent
            self.h, self.w = h, w
        if self.kind in ("full", "3d", "tail"):
            if self.kind != "full":
                self.feat = self.act(N * h * w, m.num_3d_features)
            yq, bnq = self._build_3d(B, S, h, w, self.feat)
            self.yq, self.bnq = yq, bnq
        if self.kind in ("full", "head", "tail"):
            if self.kind == "head":
                self.yq, self.bnq = self.act(N * h * w, m.num_features // S), None
            self._build_head(B, S, h, w, self.yq, self.bnq)
        if self.kind == "3d":   # separately-called forward_3d returns the activated projection
            self.out3d = self.act(N * h * w, m.num_features // S)
            self.op("f3d", "bn_res", dtype=self.code, M=N * h * w, C=m.num_features // S, y=self.yq,
                    scale=self.bnq.scale, shift=self.bnq.shift, act=1, mask=None, rows_per_group=0,
                    shortcut=None, out=self.out3d)
        if self.need_grad:
            chain = [(rec, bseg) for name, bseg in (("head", "bhead"), ("3d", "b3d"), ("2d", "b2d")) for rec in reversed(self._recs[name])]
            gout = None
            # the sub-forwards called on their own under autograd (reference: multidim_stacker.py:210-237 are ordinary
            # differentiable methods): the incoming gradient is copied into a channels-last buffer that starts the chain
            if self.kind == "3d":        # gradient wrt the ACTIVATED projection output (b, S, h, w, 256)
                self.uq_in = self.act(N * h * w, m.num_features // S)
                gout = Grad(self.uq_in)
            if self.kind == "2d":        # gradient wrt the activated 2D features (b, S, h, w, 192)
                self.dfeat_in = self.act(N * h * w, m.num_3d_features)
                gout = Grad(self.dfeat_in)
            # Gradient buckets for data parallelism: closures run in reverse parameter order, so once closure i
            # has been issued every parameter at or above min(lo of closures 0..i) is final; a cut after ~1.5 M elements
            # lets the all-reduce of that slice of the flat arena start while the rest of the backward still runs.
            self.cuts = {}           # (segment, number of ops issued) -> (lo, hi) slice of the gradient arena
            hi, lo, executed = self.grad_arena.numel, self.grad_arena.numel, []
            def flush_se(bseg):     # the deferred squeeze-excite parameter gradients of the closures issued so far: one launch
                if self._se_pending:
                    self.op(bseg, "se_fc_bwd_params_table", _struct="mds_se_fc_bwd_table_args", _jobs=[kw for _, kw in self._se_pending])
                    self._se_pending = []
            for i, (rec, bseg) in enumerate(chain):
                nxt = chain[i + 1][0] if i + 1 < len(chain) else None
                # `head` of the next closure: the BatchNorm whose backward consumes this closure's output directly
                gout = rec(bseg, gout, getattr(nxt, "head", None) if self.fuse_bn_bwd else None)
                last = gout is None or nxt is None
                lo = min(lo, getattr(rec, "lo", lo))
                cut = len(self.segs[bseg]) and (hi - lo >= self.BUCKET_ELEMS or last) and hi > lo
                if cut or last or chain[i + 1][1] != bseg:
                    flush_se(bseg)          # (a bucket is final only with every launch that adds into it issued)
                if cut:
                    self.cuts[(bseg, len(self.segs[bseg]))] = (lo, hi)
                    hi = lo
                if gout is None:
                    break
            self.grad_lo = hi        # parameters below this offset receive no gradient in this plan (frozen encoder)
            self.dfeat = gout.buf if (self.kind in ("tail", "3d") and gout is not None) else None   # gradient wrt the (b,S,h,w,192) features
            self.dyq_out = gout.buf if (self.kind == "head" and gout is not None) else None      # gradient wrt forward_head's input

    # -- helpers emitting a conv + its BN finalize
    def _pw(self, seg, x, M, K, N_, wparam, pro=None, stats_bn=None, residual=None, wt=None, epi_mode=None):
        """epi_mode (inference plans only): the output is stored as act(bn(y)) (+ residual), mds_epi_t"""
        y = self.act(M, N_)
        w = wt if wt is not None else self.pack(wparam, cabi.MDS_PACK_OI, N_, K, 1)
        # the K-heavy narrow-N launches read their filter in MFMA-fragment order, straight into registers (k_pwk8.hip)
        wfrag = (self.pack(wparam, cabi.MDS_PACK_FRAG_OI, N_, K, 1)
                 if wt is None and epi_mode is None and self.lib.fn["pw_fwd_wants_frag"](int(M), int(K), int(N_), int(self.code), 0) else None)
        if epi_mode is not None:
            assert self.eval_epilogues and stats_bn is not None
            stats_bn.finalize(self, seg)      # eval table
            self.op(seg, "pw_fwd", dtype=self.code, M=M, K=K, N=N_, x=x, w=w, y=y, pro=pro or dict(mode=0), residual=residual,
                    stats=None, epi=dict(_struct="mds_epi_t", mode=epi_mode, scale=stats_bn.scale, shift=stats_bn.shift), **self._split(M, K, N_))
            return y
        self.op(seg, "pw_fwd", dtype=self.code, M=M, K=K, N=N_, x=x, w=w, y=y, pro=pro or dict(mode=0),
                residual=residual, stats=stats_bn.stats if stats_bn is not None else None, w_frag=wfrag, form=1)
        if stats_bn is not None:
            stats_bn.finalize(self, seg)
        return y

    def _conv(self, seg, x, pro, N, IH, IW, Cin, Cout, stride, wparam, bn_mod):
        """inference plans: the output is stored as silu(bn(y)) (mds_epi_t); consumers then read it without a prologue"""
        OH, OW, pt, pl = geo.conv_geometry(IH, IW, stride)
        bn = BNL(self, bn_mod, Cout, N * OH * OW)
        dy, dx, wi = geo.taps_fwd(pt, pl)
        y = self.act(N * OH * OW, Cout)
        w = self.pack(wparam, cabi.MDS_PACK_OI, Cout, Cin, 9)
        extra = dict(stats=bn.stats)
        if self.eval_epilogues:
            extra = dict(stats=None, epi=dict(_struct="mds_epi_t", mode=EPI_BN_SILU, scale=bn.scale, shift=bn.shift))
        self.op(seg, "conv_fwd", dtype=self.code, N=N, IH=IH, IW=IW, Cin=Cin, OH=OH, OW=OW, Cout=Cout, A=OH, B=OW,
                oy0=0, ox0=0, os=1, **{"is": stride}, ntaps=9, dy=dy, dx=dx, wi=wi, wtaps=9, x=x, w=w, y=y,
                pro=pro or dict(mode=0), residual=None, **extra)
        bn.finalize(self, seg)
        return y, bn, OH, OW, (pt, pl)

    def _conv_dgrad(self, seg, dyb, N, IH, IW, Cin, Cout, stride, wparam, pads, residual, head=None):
        """grad wrt the conv input ([N][IH][IW][Cin]) from dy ([N][OH][OW][Cout]).  `head`: the BatchNorm backward that consumes
        this gradient directly (the block below's last BatchNorm) - where the launch goes to a kernel that implements
        mds_poststat_t (k_c3.hip: mds_conv_dgrad_post_ok) its sums CAN be taken in the epilogue and the Grad says so.
        MDS_FUSE_CONV_POST=1 switches that on (four bn_bwd_reduce launches and 0.75 GB per step less, parity-green) - measured
        0.05 - 0.1 ms per step SLOWER than the separate reduce passes (round 6, profiles/LOG.md), so the default leaves it off."""
        OH, OW, _, _ = geo.conv_geometry(IH, IW, stride)
        w = self.pack(wparam, cabi.MDS_PACK_IO_FLIP, Cout, Cin, 9)
        dxb = self.act(N * IH * IW, Cin)
        common = dict(dtype=self.code, N=N, IH=OH, IW=OW, Cin=Cout, OH=IH, OW=IW, Cout=Cin, wtaps=9, x=dyb, w=w,
                      y=dxb, pro=dict(mode=0), residual=residual, stats=None)
        fused = None
        if (head is not None and (head["mode"] != POST_MASK or head["rpg"] == IH * IW)
                and (head["mode"] != POST_SILU or os.environ.get("MDS_FUSE_CONV_POST_SILU", "1") == "1")
                and os.environ.get("MDS_FUSE_CONV_POST", "0") == "1"
                and self.lib.fn["conv_dgrad_post_ok"](int(self.code), int(N), int(IH), int(IW), int(Cin), int(Cout), int(stride), int(residual is not None))):
            common["post"] = head["bn"].post(head)
            fused = head["bn"]
        if stride == 1:
            dy, dx, wi = geo.taps_dgrad_s1()
            self.op(seg, "conv_fwd", A=IH, B=IW, oy0=0, ox0=0, os=1, **{"is": 1}, ntaps=9, dy=dy, dx=dx, wi=wi, **common)
        else:
            assert residual is None
            par = []
            for py in range(2):
                for px in range(2):
                    dy, dx, wi = geo.taps_dgrad_s2(py, px, pads[0], pads[1])
                    A, B_ = (IH - py + 1) // 2, (IW - px + 1) // 2
                    assert dy and A > 0 and B_ > 0
                    par.append((py, px, dy, dx, wi, A, B_))
            if Cout % 32 == 0:       # the four parities as tap groups of ONE launch: dy is read once
                self.op(seg, "conv_fwd", A=max(p[5] for p in par), B=max(p[6] for p in par), oy0=0, ox0=0, os=2, **{"is": 1},
                        ntaps=sum(len(p[2]) for p in par), dy=sum((p[2] for p in par), []), dx=sum((p[3] for p in par), []),
                        wi=sum((p[4] for p in par), []), ngroups=4, g_ntaps=[len(p[2]) for p in par],
                        g_oy0=[p[0] for p in par], g_ox0=[p[1] for p in par], g_A=[p[5] for p in par], g_B=[p[6] for p in par],
                        **common)
            else:
                assert fused is None
                for (py, px, dy, dx, wi, A, B_) in par:
                    self.op(seg, "conv_fwd", A=A, B=B_, oy0=py, ox0=px, os=2, **{"is": 1}, ntaps=len(dy), dy=dy, dx=dx,
                            wi=wi, **common)
        return Grad(dxb, fused)

    def _pw_bwd(self, seg, xin, pro, M, K, N_, wparam, dy, need_dx, residual=None, frozen=False, head=None):
        """wgrad (+ dgrad) of a 1x1 conv y[M][N] = pro(x)[M][K] w^T from the materialised dy; `head`: take the next BatchNorm
        backward's sums in the dgrad epilogue."""
        if not frozen:
            self.op(seg, "pw_wgrad", dtype=self.code, M=M, K=K, N=N_, x=xin, dy=dy, dw=self.grad(wparam), pro=pro or dict(mode=0))
        if not need_dx:
            return None
        wt = self.pack(wparam, cabi.MDS_PACK_IO_FLIP, N_, K, 1)       # [K][N]
        dx = self.act(M, K)
        extra = {}
        if head is not None:
            extra["post"] = head["bn"].post(head)
        if self.lib.fn["pw_fwd_wants_frag"](int(M), int(N_), int(K), int(self.code), 1):
            extra["w_frag"] = self.pack(wparam, cabi.MDS_PACK_FRAG_IO, N_, K, 1)
        self.op(seg, "pw_fwd", dtype=self.code, M=M, K=N_, N=K, x=dy, w=wt, y=dx, pro=dict(mode=0), residual=residual, stats=None, form=2, **extra)
        return Grad(dx, head["bn"] if head is not None else None)

    def _ir_block_eval(self, fseg, blk, bn1, bn2, bn3, xin, N, T, IH, IW, OH, OW, pt, pl, stride, groups, has_skip, kt):
        """Inference form of the inverted-residual block: the BatchNorm coefficients are known before any
        producer runs, so every producer stores its ACTIVATED output (mds_epi_t) - no BN+SiLU prologue in the consumers
        (the depthwise kernel re-evaluated it 1.33x per element), no activation copy in the pooling pass, no bn_res launch."""
        cin, mid, cout = blk.cin, blk.mid, blk.cout
        Min, Mout = N * T * IH * IW, N * T * OH * OW
        rpg = Mout // groups
        for bn in (bn1, bn2, bn3):
            bn.finalize(self, fseg)         # -> the plan's eval-BatchNorm table (one launch per forward)
        def epi(bn, mode):
            return dict(_struct="mds_epi_t", mode=mode, scale=bn.scale, shift=bn.shift)
        a1 = self.act(Min, mid)
        self.op(fseg, "pw_fwd", dtype=self.code, M=Min, K=cin, N=mid, x=xin, w=self.pack(blk.conv_pw.weight, cabi.MDS_PACK_OI, mid, cin, 1),
                y=a1, pro=dict(mode=0), residual=None, stats=None, epi=epi(bn1, EPI_BN_SILU), **self._split(Min, cin, mid))
        a2 = self.act(Mout, mid)
        R = blk.se.rd
        pooled, hidden, gate = self.zero_fwd64(groups * mid), self.f32(groups * R), self.f32(groups * mid)
        # the depthwise pass stores the activation AND takes its per-image channel means (the squeeze-excite input): no se_pool launch
        fuse_pool = groups == N and os.environ.get("MDS_EVAL_POOL", "1") == "1"
        self.op(fseg, "dw_fwd", dtype=self.code, N=N, T=T, IH=IH, IW=IW, C=mid, OH=OH, OW=OW, stride=stride, pad_t=pt,
                pad_l=pl, kt=kt, x=a1, w=P(blk.conv_dw.weight), y=a2, pro=dict(mode=0), stats=None, epi=epi(bn2, EPI_BN_SILU),
                pool=pooled if fuse_pool else None, pool_inv=1.0 / rpg if fuse_pool else 0.0)
        if not fuse_pool:
            self.op(fseg, "se_pool", dtype=self.code, groups=groups, rows_per_group=rpg, C=mid, y=a2, scale=None, shift=None,
                    pooled=pooled, act=None)
        se = blk.se
        w2t = self.pack(se.conv_expand.weight, cabi.MDS_PACK_IO_F32, mid, R, 1)
        self.op(fseg, "se_fc_fwd", groups=groups, C=mid, R=R, pooled=pooled, w1=P(se.conv_reduce.weight),
                b1=P(se.conv_reduce.bias), w2=P(se.conv_expand.weight), b2=P(se.conv_expand.bias), hidden=hidden, gate=gate,
                w2t=w2t)
        xout = self.act(Mout, cout)
        self.op(fseg, "pw_fwd", dtype=self.code, M=Mout, K=mid, N=cout, x=a2,
                w=self.pack(blk.conv_pwl.weight, cabi.MDS_PACK_OI, cout, mid, 1), y=xout,
                pro=dict(mode=PRO_GATE, scale=None, shift=None, gate=gate, rows_per_group=rpg),
                residual=xin if has_skip else None, stats=None, epi=epi(bn3, EPI_AFFINE), **self._split(Mout, mid, cout))
        return xout, OH, OW

    # -- inverted-residual block (2D: T=1, kt=1 ; 3D: kt=3), shared by encoder stages 3-5 and conv3d_encoder
    def _ir_block(self, fseg, recs, blk, bn1m, bn2m, bn3m, xin, N, T, IH, IW, stride, groups, has_skip, frozen):
        cin, mid, cout = blk.cin, blk.mid, blk.cout
        kt = 3 if isinstance(blk.conv_dw, torch.nn.Conv3d) else 1
        OH, OW, pt, pl = geo.conv_geometry(IH, IW, stride)
        Min, Mout = N * T * IH * IW, N * T * OH * OW
        rpg = Mout // groups
        bn1, bn2, bn3 = BNL(self, bn1m, mid, Min), BNL(self, bn2m, mid, Mout), BNL(self, bn3m, cout, Mout)
        if self.eval_epilogues and (kt == 1 or T == 5):
            return self._ir_block_eval(fseg, blk, bn1, bn2, bn3, xin, N, T, IH, IW, OH, OW, pt, pl, stride, groups, has_skip, kt)
        y1 = self._pw(fseg, xin, Min, cin, mid, blk.conv_pw.weight, stats_bn=bn1)
        y2 = self.act(Mout, mid)
        wdw = P(blk.conv_dw.weight)
        self.op(fseg, "dw_fwd", dtype=self.code, N=N, T=T, IH=IH, IW=IW, C=mid, OH=OH, OW=OW, stride=stride, pad_t=pt,
                pad_l=pl, kt=kt, x=y1, w=wdw, y=y2, pro=bn1.pro(), stats=bn2.stats)
        bn2.finalize(self, fseg)
        R = blk.se.rd
        pooled, hidden, gate = self.zero_fwd64(groups * mid), self.f32(groups * R), self.f32(groups * mid)
        # The projection and its weight gradient read y2 through the BN + SiLU + gate prologue; the pooling pass only reads.
        # (Rounds 2-3 had the pooling pass also WRITE silu(bn2(y2)) for them - free while that pass was bound by its atomics'
        # epilogue; with the per-channel epilogue the 2 x mid-width write + read costs more than the prologue's arithmetic:
        # 12.84 -> 12.68 ms per step, MDS_SE_ACT=1 restores it.)
        keep_act = os.environ.get("MDS_SE_ACT", "0") == "1"
        a2 = self.act(Mout, mid) if keep_act else y2
        self.op(fseg, "se_pool", dtype=self.code, groups=groups, rows_per_group=rpg, C=mid, y=y2, scale=bn2.scale,
                shift=bn2.shift, pooled=pooled, act=a2 if keep_act else None)
        gate_pro = (dict(mode=PRO_GATE, scale=None, shift=None, gate=gate, rows_per_group=rpg) if keep_act else
                    dict(mode=PRO_BN_GATE, scale=bn2.scale, shift=bn2.shift, gate=gate, rows_per_group=rpg))
        se = blk.se
        w2t = self.pack(se.conv_expand.weight, cabi.MDS_PACK_IO_F32, mid, R, 1)    # [R][mid], fp32
        self.op(fseg, "se_fc_fwd", groups=groups, C=mid, R=R, pooled=pooled, w1=P(se.conv_reduce.weight),
                b1=P(se.conv_reduce.bias), w2=P(se.conv_expand.weight), b2=P(se.conv_expand.bias), hidden=hidden, gate=gate,
                w2t=w2t)
        y3 = self._pw(fseg, a2, Mout, mid, cout, blk.conv_pwl.weight, pro=gate_pro, stats_bn=bn3)
        mask = self.mask(groups, blk.dpr) if has_skip else None
        xout = self.act(Mout, cout)
        self.op(fseg, "bn_res", dtype=self.code, M=Mout, C=cout, y=y3, scale=bn3.scale, shift=bn3.shift, act=0, mask=mask,
                rows_per_group=rpg, shortcut=xin if has_skip else None, out=xout)
        self.taps.append(dict(tag=f"{fseg}.ir{len(self.taps)}", buf=xout, bn=None, rows=Mout, C=cout))

        def bwd(seg, dout, nxt_head):
            g3 = gsrc(G_MASK, dout.buf, mask=mask, rpg=rpg) if mask is not None else gsrc(G_PLAIN, dout.buf)
            dy3 = self.act(Mout, cout)      # (the sums may have been taken by the producer of dout, a 1x1 data-gradient GEMM's epilogue)
            bn3.backward(self, seg, g3, y3, dy3, reduce=dout.reduced is not bn3, frozen=frozen)
            u2 = self._pw_bwd(seg, a2, gate_pro, Mout, mid, cout, blk.conv_pwl.weight, dy3, True, frozen=frozen).buf
            dgate, dpool = self.zero_bwd64(groups * mid), self.f32(groups * mid)
            nblk = self.lib.fn["se_bwd_reduce_blocks"](rpg, mid)
            bnsums = self.f32(groups * nblk * 4 * mid)
            self.op(seg, "se_bwd_reduce", dtype=self.code, groups=groups, rows_per_group=rpg, C=mid, u=u2, y=y2,
                    scale=bn2.scale, shift=bn2.shift, dgate=dgate, mean=bn2.mean, rstd=bn2.rstd, bnsums=bnsums)
            if frozen:
                sg = [self.f32(se.conv_reduce.weight.numel()), self.f32(R), self.f32(se.conv_expand.weight.numel()), self.f32(mid)]
            else:
                sg = [self.grad(se.conv_reduce.weight), self.grad(se.conv_reduce.bias), self.grad(se.conv_expand.weight),
                      self.grad(se.conv_expand.bias)]
            sekw = dict(groups=groups, C=mid, R=R, rows_per_group=rpg, dgate=dgate, gate=gate, hidden=hidden,
                        pooled=pooled, w1=P(se.conv_reduce.weight), w2=P(se.conv_expand.weight), dpooled=dpool,
                        scratch=self.f32(groups * R), dw1=sg[0], db1=sg[1], dw2=sg[2], db2=sg[3],
                        bnsums=bnsums, bn_nblk=nblk, bn_stats=bn2.bstats, w2t=w2t)
            self.op(seg, "se_fc_bwd_data", _struct="mds_se_fc_bwd_args", **sekw)
            if self.se_table:           # parameter gradients: leaves - deferred to the end of the gradient bucket, one table launch (second stream)
                self._se_pending.append((seg, sekw))
            else:
                self.op(seg, "se_fc_bwd_params", _struct="mds_se_fc_bwd_args", **sekw)
            dy2 = self.act(Mout, mid)
            bn2.backward(self, seg, gsrc(G_SE, u2, gate=gate, dpooled=dpool, rpg=rpg), y2, dy2, reduce=False, frozen=frozen)
            g1 = self.act(Min, mid)
            self.op(seg, "dw_bwd", dtype=self.code, N=N, T=T, IH=IH, IW=IW, C=mid, OH=OH, OW=OW, stride=stride, pad_t=pt,
                    pad_l=pl, kt=kt, x=y1, dy=dy2, w=wdw, g=g1, dw=self.f32(mid * kt * 9) if frozen else self.grad(blk.conv_dw.weight),
                    pro=bn1.pro(), mean=bn1.mean, rstd=bn1.rstd, stats=bn1.bstats)
            dy1 = self.act(Min, mid)
            bn1.backward(self, seg, gsrc(G_PLAIN, g1), y1, dy1, reduce=False, frozen=frozen)
            return self._pw_bwd(seg, xin, None, Min, cin, mid, blk.conv_pw.weight, dy1, True,
                                residual=dout.buf if has_skip else None, frozen=frozen, head=nxt_head)

        bwd.head = bn3.head(y3, POST_MASK if mask is not None else POST_PLAIN, mask, rpg)
        bwd.lo = self._lo(blk)
        recs.append(bwd)
        return xout, OH, OW

    # -- 2D: stem + 21 encoder blocks + conv2d_projection  (forward_2d, :210-219)
    def _build_2d(self, N, H, W):
        m, enc = self.m, self.m.conv2d_encoder
        recs = self._recs["2d"]
        fr = not self.enc_grad
        OH, OW, pt, pl = geo.conv_geometry(H, W, 2)
        y0 = self.act(N * OH * OW, 32)
        bn0 = BNL(self, enc.bn1, 32, N * OH * OW)
        wst = self.pack(enc.conv_stem.weight, cabi.MDS_PACK_STEM, 32, 27, 1)
        extra = {}
        if self.ingest is not None:      # raw frames: [nsrc][3][src_h][src_w] uint8, images beyond nsrc are the mirrored TTA copies
            src_h, src_w, nsrc = self.ingest
            assert not self.need_grad and N in (nsrc, 2 * nsrc) and src_h <= H and src_w <= W
            self.x_u8 = self._own(nsrc * 3 * src_h * src_w, torch.uint8)
            extra["ingest"] = dict(_struct="mds_ingest_t", u8=self.x_u8, nsrc=nsrc, src_h=src_h, src_w=src_w,
                                   pad_top=(H - src_h) // 2, pad_left=(W - src_w) // 2, scale=1.0 / 255.0)
        if self.eval_epilogues:
            extra.update(stats=None, epi=dict(_struct="mds_epi_t", mode=EPI_BN_SILU, scale=bn0.scale, shift=bn0.shift))
        else:
            extra.update(stats=bn0.stats)
        self.op("f2d", "stem_fwd", dtype=self.code, N=N, H=H, W=W, OH=OH, OW=OW, Cout=32, pad_t=pt, pad_l=pl,
                x=None if self.ingest is not None else self.x_in, w=wst, y=y0, **extra)
        bn0.finalize(self, "f2d")

        def stem_bwd(seg, u0, nxt_head):
            if fr:
                return None
            # (u0.reduced is bn0: the first 3x3 layer's data gradient stored g = u * silu'(z) and took the sums - k_c3.hip, POST_SILU)
            g0 = gsrc(G_PLAIN, u0.buf) if u0.reduced is bn0 else gsrc(G_SILU, u0.buf)
            if self.tdt == torch.bfloat16 and os.environ.get("MDS_STEM_DYP", "1") == "1":
                # the stem has no data gradient: its BatchNorm-backward apply pass would only feed the weight gradient, which
                # forms dy = A*u*silu'(z) + B*y + D on load instead (no 0.9 GB apply launch at the very end of the step)
                if u0.reduced is not bn0:
                    bn0.bwd_reduce(self, seg, g0, y0)
                bn0.bwd_finalize(self, seg)
                self.op(seg, "stem_wgrad", dtype=self.code, N=N, H=H, W=W, OH=OH, OW=OW, Cout=32, pad_t=pt, pad_l=pl,
                        x=self.x_in, dy=None, dw=self.grad(enc.conv_stem.weight),
                        dyp=dict(_struct="mds_dyp_t", mode=1, g=g0, y=y0, bn=bn0.buf, lin=bn0.lin))
                return None
            dy0 = self.act(N * OH * OW, 32)
            bn0.backward(self, seg, g0, y0, dy0, reduce=u0.reduced is not bn0)
            self.op(seg, "stem_wgrad", dtype=self.code, N=N, H=H, W=W, OH=OH, OW=OW, Cout=32, pad_t=pt, pad_l=pl,
                    x=self.x_in, dy=dy0, dw=self.grad(enc.conv_stem.weight))
            return None

        stem_bwd.head = bn0.head(y0, POST_SILU)
        stem_bwd.lo = self._lo(enc.conv_stem, enc.bn1)
        recs.append(stem_bwd)
        cur, cur_bn, ch, cw = y0, (None if self.eval_epilogues else bn0), OH, OW      # cur_bn != None: `cur` is a raw tensor read through BN+SiLU
        for blk in enc.block_list():
            if blk.kind == "cn":
                cur, cur_bn, ch, cw = self._cn_block(recs, blk, cur, cur_bn, N, ch, cw, fr)
            elif blk.kind == "er":
                cur, ch, cw = self._er_block(recs, blk, cur, cur_bn, N, ch, cw, fr)
                cur_bn = None
            else:
                assert cur_bn is None
                cur, ch, cw = self._ir_block("f2d", recs, blk, blk.bn1, blk.bn2, blk.bn3, cur, N, 1, ch, cw, blk.stride,
                                             N, blk.has_skip, fr)
        # conv2d_projection: 1x1 + BN + SiLU, materialised (it is block 0's input AND shortcut)
        M, cf = N * ch * cw, m.num_3d_features
        cenc = enc.feature_info[-1]["num_chs"]
        bnp = BNL(self, m.conv2d_projection[1], cf, M)
        if self.eval_epilogues:          # inference: BN + SiLU in the projection's epilogue
            feat = self._pw("f2d", cur, M, cenc, cf, m.conv2d_projection[0].weight, stats_bn=bnp, epi_mode=EPI_BN_SILU)
            return feat, ch, cw
        yp = self._pw("f2d", cur, M, cenc, cf, m.conv2d_projection[0].weight, stats_bn=bnp)
        feat = self.act(M, cf)
        self.op("f2d", "bn_res", dtype=self.code, M=M, C=cf, y=yp, scale=bnp.scale, shift=bnp.shift, act=1, mask=None,
                rows_per_group=0, shortcut=None, out=feat)
        self.taps.append(dict(tag="f2d.proj", buf=feat, bn=None, rows=M, C=cf))
        xenc = cur

        def proj_bwd(seg, dfeat, nxt_head):
            dyp = self.act(M, cf)
            if dfeat.reduced is bnp:      # the 3D tail's last data-gradient GEMM stored g = dfeat*silu'(z) and took the sums
                bnp.backward(self, seg, gsrc(G_PLAIN, dfeat.buf), yp, dyp, reduce=False)
            else:
                bnp.backward(self, seg, gsrc(G_SILU, dfeat.buf), yp, dyp)
            return self._pw_bwd(seg, xenc, None, M, cenc, cf, m.conv2d_projection[0].weight, dyp, need_dx=not fr, head=nxt_head)

        proj_bwd.head = bnp.head(yp, POST_SILU)
        proj_bwd.lo = self._lo(m.conv2d_projection)
        recs.append(proj_bwd)
        return feat, ch, cw

    def _cn_block(self, recs, blk, xin, xin_bn, N, IH, IW, fr):
        assert not blk.has_skip and (xin_bn is not None or self.eval_epilogues)
        y, bn1, OH, OW, pads = self._conv("f2d", xin, xin_bn.pro() if xin_bn is not None else None, N, IH, IW, blk.cin, blk.cout,
                                          blk.stride, blk.conv.weight, blk.bn1)
        if self.eval_epilogues:
            return y, None, OH, OW       # activated output, no backward
        self.taps.append(dict(tag=f"f2d.cn{len(self.taps)}", buf=y, bn=bn1, rows=N * OH * OW, C=blk.cout))

        def bwd(seg, u, nxt_head):
            if fr:
                return None
            dy_ = self.act(N * OH * OW, blk.cout)
            if u.reduced is bn1:      # the producer (a 3x3 data gradient of k_c3.hip) stored g = u * silu'(z) and took the sums
                bn1.backward(self, seg, gsrc(G_PLAIN, u.buf), y, dy_, reduce=False)
            else:
                bn1.backward(self, seg, gsrc(G_SILU, u.buf), y, dy_)
            self._conv_wgrad(seg, xin, xin_bn.pro(), N, IH, IW, blk.cin, OH, OW, blk.cout, blk.stride, pads, dy_, blk.conv.weight)
            return self._conv_dgrad(seg, dy_, N, IH, IW, blk.cin, blk.cout, blk.stride, blk.conv.weight, pads, None, head=nxt_head)

        bwd.head = bn1.head(y, POST_SILU)
        bwd.lo = self._lo(blk)
        recs.append(bwd)
        return y, bn1, OH, OW

    def _conv_wgrad(self, seg, x, pro, N, IH, IW, Cin, OH, OW, Cout, stride, pads, dyb, wparam):
        dy, dx, wi = geo.taps_fwd(pads[0], pads[1])
        self.op(seg, "conv_wgrad", dtype=self.code, N=N, IH=IH, IW=IW, Cin=Cin, OH=OH, OW=OW, Cout=Cout,
                **{"is": stride}, ntaps=9, dy=dy, dx=dx, wi=wi, wtaps=9, x=x, dyt=dyb, dw=self.grad(wparam),
                pro=pro or dict(mode=0))

    def _er_block(self, recs, blk, xin, xin_bn, N, IH, IW, fr):
        cin, mid, cout = blk.cin, blk.mid, blk.cout
        pro_in = xin_bn.pro() if xin_bn is not None else None
        ya, bn1, OH, OW, pads = self._conv("f2d", xin, pro_in, N, IH, IW, cin, mid, blk.stride, blk.conv_exp.weight, blk.bn1)
        M = N * OH * OW
        bn2 = BNL(self, blk.bn2, cout, M)
        has_skip = blk.has_skip
        assert not has_skip or xin_bn is None
        if self.eval_epilogues:          # inference: BN2 + shortcut in the projection's epilogue, no bn_res launch
            xout = self._pw("f2d", ya, M, mid, cout, blk.conv_pwl.weight, pro=None, stats_bn=bn2,      # ya = silu(bn1(.)) already
                            residual=xin if has_skip else None, epi_mode=EPI_AFFINE)
            return xout, OH, OW       # (inference plans have no backward closures)
        yb = self._pw("f2d", ya, M, mid, cout, blk.conv_pwl.weight, pro=bn1.pro(), stats_bn=bn2)
        mask = self.mask(N, blk.dpr) if has_skip else None
        rpg = OH * OW
        xout = self.act(M, cout)
        self.op("f2d", "bn_res", dtype=self.code, M=M, C=cout, y=yb, scale=bn2.scale, shift=bn2.shift, act=0, mask=mask,
                rows_per_group=rpg, shortcut=xin if has_skip else None, out=xout)
        self.taps.append(dict(tag=f"f2d.er{len(self.taps)}", buf=xout, bn=None, rows=M, C=cout))

        def bwd(seg, dout, nxt_head):
            if fr:
                return None
            g2 = gsrc(G_MASK, dout.buf, mask=mask, rpg=rpg) if mask is not None else gsrc(G_PLAIN, dout.buf)
            dya = self.act(M, mid)
            dyb = self.act(M, cout)
            bn2.backward(self, seg, g2, yb, dyb, reduce=dout.reduced is not bn2)
            ua = self._pw_bwd(seg, ya, bn1.pro(), M, mid, cout, blk.conv_pwl.weight, dyb, True)
            bn1.backward(self, seg, gsrc(G_SILU, ua.buf), ya, dya)
            self._conv_wgrad(seg, xin, pro_in, N, IH, IW, cin, OH, OW, mid, blk.stride, pads, dya, blk.conv_exp.weight)
            return self._conv_dgrad(seg, dya, N, IH, IW, cin, mid, blk.stride, blk.conv_exp.weight, pads,
                                    dout.buf if has_skip else None, head=nxt_head)

        bwd.head = bn2.head(yb, POST_MASK if mask is not None else POST_PLAIN, mask, rpg)
        bwd.lo = self._lo(blk)
        recs.append(bwd)
        return xout, OH, OW

    # -- 3D: 4 inverted-residual blocks + conv3d_projection  (forward_3d, :221-230).  Channels-last
    #    rows [b][t][h][w][c] make both transposes of the reference disappear.
    def _build_3d(self, B, S, h, w, feat):
        m = self.m
        recs = self._recs["3d"]
        cur = feat
        for blk in m.conv3d_encoder:
            cur, _, _ = self._ir_block("f3d", recs, blk, blk.bn1.bn3d, blk.bn2.bn3d, blk.bn3.bn3d, cur, B, S, h, w, 1, B,
                                       True, False)
        M, cf, cq = B * S * h * w, m.num_3d_features, m.num_features // S
        bnq = BNL(self, m.conv3d_projection[1], cq, M)
        yq = self._pw("f3d", cur, M, cf, cq, m.conv3d_projection[0].weight, stats_bn=bnq)
        self.taps.append(dict(tag="f3d.proj", buf=yq, bn=bnq, rows=M, C=cq))
        x3 = cur

        def bwd(seg, uq, nxt_head):
            dyq = self.act(M, cq)
            bnq.backward(self, seg, gsrc(G_SILU, uq.buf), yq, dyq)
            return self._pw_bwd(seg, x3, None, M, cf, cq, m.conv3d_projection[0].weight, dyq, True, head=nxt_head)

        bwd.lo = self._lo(m.conv3d_projection)
        recs.append(bwd)
        return yq, bnq

    # -- head: GeM over (h, w) for every (b, t, c) + dropout + Linear  (forward_head, :232-237)
    def _build_head(self, B, S, h, w, yq, bnq):
        m = self.m
        cq = m.num_features // S
        F_ = S * cq
        pro = bnq.pro() if bnq is not None else dict(mode=0)
        gp = m.global_pool
        pooled = self.f32(B * F_)
        self.op("fhead", "gem_fwd", dtype=self.code, groups=B * S, rows_per_group=h * w, C=cq, y=yq, pro=pro, p=P(gp.p),
                eps=float(gp.eps), pooled=pooled, accum=self.zero_fwd64(B * S * cq))
        dmask = self.mask(B * F_, m.drop_rate) if m.drop_rate > 0 else None
        ncls = m.classifier.out_features
        self.logits = self.f32(B * ncls)
        # the predictor's tail plans (ingest = ("probs", tta)): nn.Sigmoid + the mean over the TTA pair in the head's launch
        tta = self.ingest[1] if (self.kind == "tail" and self.ingest is not None and self.ingest[0] == "probs") else 0
        self.probs = self.f32(B // tta * ncls) if tta else None
        self.op("fhead", "head_fwd", B=B, F=F_, NC=ncls, pooled=pooled, mask=dmask, w=P(m.classifier.weight),
                b=P(m.classifier.bias), logits=self.logits, probs=self.probs, tta=max(tta, 1))
        self.dlogits = self.f32(B * ncls) if self.need_grad else None

        def bwd(seg, _, nxt_head):
            dpo = self.f32(B * F_)
            self.op(seg, "head_bwd", B=B, F=F_, NC=ncls, pooled=pooled, mask=dmask, w=P(m.classifier.weight),
                    dlogits=self.dlogits, dpooled=dpo, dw=self.grad(m.classifier.weight), db=self.grad(m.classifier.bias))
            uq = self.act(B * S * h * w, cq)
            self.op(seg, "gem_bwd", dtype=self.code, groups=B * S, rows_per_group=h * w, C=cq, y=yq, pro=pro, p=P(gp.p),
                    eps=float(gp.eps), pooled=pooled, dpooled=dpo, u=uq, dp=self.grad(gp.p), accum=self.zero_bwd64(B * S * cq))
            return Grad(uq)

        bwd.lo = self._lo(gp, m.classifier)
        self._recs["head"].append(bwd)

    # ------------------------------------------------------------------ binding
    def _finalize(self):
        dev = self.device
        # deferred squeeze-excite parameter gradients that no gradient-bucket cut flushed (a builder that called a block's backward
        # closure on its own): they close the segment they were recorded in
        for seg in dict.fromkeys(sg for sg, _ in self._se_pending):
            self.op(seg, "se_fc_bwd_params_table", _struct="mds_se_fc_bwd_table_args", _jobs=[kw for sg, kw in self._se_pending if sg == seg])
        self._se_pending = []
        self.zf_arena.numel, self.zb_arena.numel, self.mask_arena.numel, self.zb64_arena.numel = self._zf, self._zb, self._mask_total, self._zb64
        self.zf64_arena.numel = self._zf64
        self.mask_arena.tensor = torch.zeros(max(self.mask_arena.numel, 1), dtype=torch.float32, device=dev)
        # everything a pass zeroes lives in ONE buffer per pass (fp64 part first: 8-byte aligned), so that begin_forward /
        # begin_backward are one memset each instead of two / three dependent launches in front of the first kernel
        def carve(*arenas):
            size = lambda a: max(a.numel, 1) * (8 if a.dtype == torch.float64 else 4)
            pad = lambda n: (n + 255) // 256 * 256           # every arena starts on a 256-byte boundary (16-byte vector accesses)
            raw = torch.zeros(sum(pad(size(a)) for a in arenas), dtype=torch.uint8, device=dev)
            off = 0
            for a in arenas:
                a.tensor = raw[off:off + size(a)].view(a.dtype)
                off += pad(size(a))
            return raw
        self._zero_fwd = carve(self.zf64_arena, self.zf_arena)
        self._zero_bwd = carve(self.zb64_arena, self.zb_arena, self.grad_arena)
        for l in self._lazy:
            l.tensor = (torch.zeros if l.kind == "own0" else torch.empty)(max(l.numel, 1), dtype=l.dtype, device=dev)
        if self.masks:
            keep = torch.empty(self._mask_total, dtype=torch.float32)
            for off, n, kp in self.masks:
                keep[off:off + n] = kp
            self.mask_keep = keep.to(dev)
        # weight-pack job table (device resident)
        Job = cabi.STRUCTS["mds_pack_job"]
        jobs = (Job * max(len(self.pack_jobs), 1))()
        self.pack_max = 1
        for j, (p, dst, kind, O, I, taps) in enumerate(self.pack_jobs):
            jobs[j].src = p.detach().data_ptr()
            jobs[j].dst = dst.resolve().data_ptr()
            jobs[j].kind, jobs[j].O, jobs[j].I, jobs[j].taps = kind, O, I, taps
            self.pack_max = max(self.pack_max, dst.numel)
        self.pack_table = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(dev)
        self.param_ptrs = tuple(p.data_ptr() for p, *_ in self.pack_jobs)
        if self.eval_bn:
            BJ = cabi.STRUCTS["mds_bn_eval_job"]
            bj = (BJ * len(self.eval_bn))()
            for j, bn_ in enumerate(self.eval_bn):
                m, C_, out = bn_.mod, bn_.C, bn_.buf
                bj[j].gamma, bj[j].beta = m.weight.detach().data_ptr(), m.bias.detach().data_ptr()
                bj[j].running_mean, bj[j].running_var = m.running_mean.data_ptr(), m.running_var.data_ptr()
                bj[j].out, bj[j].eps, bj[j].C = out.resolve().data_ptr(), float(m.eps), C_
            self.eval_bn_table = torch.frombuffer(bytearray(bytes(bj)), dtype=torch.uint8).to(dev)
            self.eval_bn_maxc = max(e.C for e in self.eval_bn)
            self.param_ptrs += tuple(t.data_ptr() for e in self.eval_bn for t in (e.mod.weight, e.mod.bias, e.mod.running_mean, e.mod.running_var))
        # bind every recorded launch
        self.bound: Dict[str, list] = {}
        self.costs: Dict[str, list] = {}
        self._keep = []
        self.input_slots = []
        for seg, ops in self.segs.items():
            out = []
            for name, kw in ops:
                self._input_fields = []
                st = self._bind(kw.get("_struct", f"mds_{name}_args"), kw)
                for _, field in self._input_fields:
                    self.input_slots.append((st, field))
                out.append((name if not kw.get("_side") else name + "@side", self.lib.fn[name], st, C.byref(st)))
            self.bound[seg] = out
            self.costs[seg] = [op_cost(name, kw, 2 if self.code == cabi.MDS_BF16 else 4) for name, kw in ops]
        self.nbytes = sum(l.tensor.numel() * l.tensor.element_size() for l in self._lazy)

    def _bind(self, struct_name, kw):
        vals = {}
        for k, v in kw.items():
            if k in ("_struct", "_side"):
                continue
            if k == "_jobs":          # a device-resident array of bound mds_se_fc_bwd_args (mds_se_fc_bwd_table_args)
                arr = (cabi.STRUCTS["mds_se_fc_bwd_args"] * len(v))()
                for j, kwj in enumerate(v):
                    arr[j] = self._bind("mds_se_fc_bwd_args", kwj)
                tab = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device)
                self._keep.append(tab)
                vals.update(jobs=tab, njobs=len(v), max_rc=max(kwj["R"] * kwj["C"] for kwj in v))
                continue
            if isinstance(v, dict):
                v = self._bind(v.get("_struct", "mds_pro_t"), v)
            elif isinstance(v, Lazy) and v.kind == "input":
                self._input_fields.append((struct_name, k))
                v = 0
            elif isinstance(v, (Lazy, P)):
                t = v.resolve()
                self._keep.append(t)
                v = t
            vals[k] = v
        return cabi.make(struct_name, **vals)

    # ------------------------------------------------------------------ execution
    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream if self.device.type == "cuda" else 0

    def device_guard(self):
        """every launch, the side stream and the mask RNG belong to the plan's device, whatever the
        caller's current device is"""
        return torch.cuda.device(self.device) if self.device.type == "cuda" else contextlib.nullcontext()

    # (the stem's weight gradient stays on the dependent chain: it is its last launch, and the second stream still has the first
    #  3x3 layer's weight gradient to finish - 14.30 vs 14.35 ms per step)
    SIDE_OPS = ("pw_wgrad", "conv_wgrad", "se_fc_bwd_params", "se_fc_bwd_params_table")
    BUCKET_ELEMS = 1_500_000

    def _lo(self, *mods_or_params):
        """lowest gradient-arena offset among the given parameters / modules' parameters"""
        ps = []
        for m in mods_or_params:
            ps.extend(m.parameters() if isinstance(m, torch.nn.Module) else [m])
        return min(self.poff[id(p)] for p in ps)

    cut_hook = None     # data parallelism: called as cut_hook(plan, lo, hi) when arena[lo:hi] is final (all its launches issued)

    def _failed(self, rc, name):
        """a launch was refused: split-K tickets reset themselves only when a launch completes - zero them so that the plan's
        later launches do not start from a stale count - then raise"""
        tk = getattr(self, "_split_ticket", None)
        if tk is not None and tk.tensor is not None:
            tk.tensor.zero_()
        self.lib.check(rc, name.split("@")[0])

    def run(self, seg):
        if self.profile is not None:
            return self._run_profiled(seg)
        stream = self._stream()
        side = self._side_stream() if seg[0] == "b" else None
        hook = self.cut_hook if seg[0] == "b" else None
        if side is None:
            for k, (name, fn, st, ref) in enumerate(self.bound[seg]):
                rc = fn(ref, stream)
                if rc:
                    self._failed(rc, name)
                if hook is not None and (seg, k + 1) in self.cuts:
                    hook(self, *self.cuts[(seg, k + 1)])
            return
        # Backward: the weight-gradient GEMMs are leaves of the dependency graph (they only add
        # into the gradient arena), so they go to a second HIP stream and fill the CUs that the
        # short dgrad / BN-backward launches of the critical path leave idle.
        # Hand-off (round 4): the second stream waits for the STOP event of the dependent chain's last kernel before the side
        # launch (mds_launch_event: the kernel's own completion signal) - an event RECORDED on the chain is a marker packet that
        # costs it 4.4 us (tools/probes/event_cost.py), 71 times per step.  MDS_SIDE_EVENTS=record restores the recorded events.
        main = torch.cuda.current_stream(self.device)
        side_h = side.cuda_stream
        ops = self.bound[seg]
        is_side = self._side_flags.get(seg)
        if is_side is None:
            is_side = self._side_flags[seg] = [name in self.SIDE_OPS or name.endswith("@side") for name, *_ in ops]
        ext = self._ext_events()
        evs, n = self._side_events, 0
        armed = None          # stop event bound to the chain's most recent kernel (None: nothing to wait for yet in this segment)
        for k, (name, fn, st, ref) in enumerate(ops):
            if is_side[k]:
                if ext is not None and armed is not None:
                    ext.wait(side_h, armed)
                else:
                    if n == len(evs):
                        evs.append(torch.cuda.Event())
                    ev = evs[n]; n += 1
                    ev.record(main)
                    side.wait_event(ev)
                rc = fn(ref, side_h)
            elif ext is not None and k + 1 < len(ops) and is_side[k + 1]:
                armed = ext.get(seg, k)
                self.lib.fn["launch_event"](armed)
                try:
                    rc = fn(ref, stream)
                finally:          # the thread-local stop event never stays armed past this launch
                    used = self.lib.fn["launch_event"](None)
                if used == 0:     # the op took a path without a kernel launch: the event still names LAST step's kernel -
                    armed = None  # fall back to a recorded event for the side launch that follows
            else:
                rc = fn(ref, stream)
            if rc:
                self._failed(rc, name)
            if hook is not None and (seg, k + 1) in self.cuts:
                hook(self, *self.cuts[(seg, k + 1)])

    class _ExtEvents:
        """hipEvent_t handles used as kernel STOP events (hipExtLaunchKernelGGL through mds_launch_event) + hipStreamWaitEvent"""

        def __init__(self, device):
            import ctypes
            self.ct = ctypes
            self.hip = ctypes.CDLL(_hip_path())
            self.hip.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
            self.hip.hipStreamWaitEvent.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
            self.device, self.ev = device, {}
            # hipEventDisableTiming | hipEventDisableSystemFence: these events only order the second stream behind a kernel of
            # the chain - nobody on the host inspects them, so the system-scope fence (cache write-back for the host's benefit)
            # a default event adds when it completes is dropped; the kernel's own device-scope release and the waiting stream's
            # acquire stay.  12.67 -> 12.56 ms per step same-box (profiles/r05_ab_event_flags.txt; MDS_EVENT_FLAGS=0 = default events).
            self.flags = int(os.environ.get("MDS_EVENT_FLAGS", "0x20000002"), 0)

        def _create(self):
            """one hipEvent_t with the cheapest flags this runtime accepts: disable-timing | disable-system-fence (ROCm >= 6.x), then
            disable-timing alone, then a default event - the first combination that works is remembered (a runtime that
            rejects the fence flag must not fail the first backward)"""
            v = self.ct.c_void_p()
            tried = []
            for flags in dict.fromkeys((self.flags, 0x2, 0x0)):
                with torch.cuda.device(self.device):
                    rc = self.hip.hipEventCreateWithFlags(self.ct.byref(v), flags)
                if rc == 0 and v.value:
                    self.flags = flags
                    return v.value
                tried.append((hex(flags), rc))
            raise RuntimeError(f"hipEventCreateWithFlags failed for every flag combination: {tried}")

        def get(self, seg, k):
            h = self.ev.get((seg, k))
            if h is None:
                h = self.ev[(seg, k)] = self._create()
            return h

        def wait(self, stream_handle, event):
            rc = self.hip.hipStreamWaitEvent(stream_handle, event, 0)
            if rc != 0:
                raise RuntimeError(f"hipStreamWaitEvent failed: {rc}")

    def _ext_events(self):
        if os.environ.get("MDS_SIDE_EVENTS", "stop") != "stop":
            return None
        if getattr(self, "_ext", None) is None:
            self._ext = Plan._ExtEvents(self.device)
        return self._ext

    def join_backward(self):
        """the gradient arena is complete once the side stream has drained"""
        if getattr(self, "_side", None) is not None and self.profile is None:
            torch.cuda.current_stream(self.device).wait_stream(self._side)

    def _side_stream(self):
        if self.device.type != "cuda" or os.environ.get("MDS_SIDE_STREAM", "1") == "0":
            return None
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=self.device)
            self._side_events = []
            self._side_flags = {}
        return self._side

    def _run_profiled(self, seg):
        """bench.py's per-kernel pass: a HIP event pair (on the launch stream) around every launch."""
        stream = self._stream()
        for (name, fn, st, ref), cost in zip(self.bound[seg], self.costs[seg]):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(ref, stream)
            e1.record()
            if rc:
                self.lib.check(rc, name)
            self.profile.append((name.split("@")[0], seg, e0, e1, cost))

    def read_taps(self):
        """[(tag, fp32 tensor [rows][C])] of every block output of the last forward (training plans materialise them; a tap
        with a BatchNorm attached is stored raw and read through BN + SiLU here) - the per-layer parity tests' view"""
        out = []
        for t in self.taps:
            v = t["buf"].resolve().view(t["rows"], t["C"]).float()
            if t["bn"] is not None:
                v = torch.nn.functional.silu(v * t["bn"].scale.resolve() + t["bn"].shift.resolve())
            out.append((t["tag"], v))
        return out

    def pack_weights(self):
        if self.pack_jobs:
            self.lib.check(self.lib.fn["pack_weights"](self.pack_table.data_ptr(), len(self.pack_jobs), self.pack_max,
                                                       self.code, self._stream()), "pack_weights")

    def bind_input(self, x):
        """point the stem kernels at the caller's (B,T,H,W) fp32 frame stack — no staging copy."""
        assert x.dtype == torch.float32 and x.is_contiguous() and x.numel() == self.x_in.numel
        ptr = x.data_ptr()
        for st, field in self.input_slots:
            setattr(st, field, ptr)
        self._x_ref = x

    def bind_dlogits(self, dlogits):
        self.dlogits.tensor.copy_(dlogits.reshape(-1).float())

    def weight_tensors(self):
        """every tensor whose CONTENT the weight-dependent prefix of a forward reads (packed weights, eval BatchNorm table)"""
        ts = [p for p, *_ in self.pack_jobs]
        for e in self.eval_bn:
            m = e.mod
            ts += [m.weight, m.bias, m.running_mean, m.running_var]
        return ts

    def refresh_weights(self):
        """the weight-dependent prefix of a forward: the eval-mode BatchNorm table and the packed filter copies"""
        if self.eval_bn:
            self.lib.check(self.lib.fn["bn_eval_table"](self.eval_bn_table.data_ptr(), len(self.eval_bn), self.eval_bn_maxc,
                                                        self._stream()), "bn_eval_table")
        self.pack_weights()

    def begin_forward(self, mask_override=None, refresh=True):
        """refresh=False (the stream predictor): the caller runs refresh_weights() itself, and only when a parameter changed"""
        if self.zf_arena.numel or self.zf64_arena.numel:
            self._memset(self._zero_fwd)
        if self.masks:
            if mask_override is not None:
                self.mask_arena.tensor.copy_(mask_override.to(self.device, torch.float32).view(-1))
            else:       # Bernoulli(keep) / keep in two launches (was five: rand, compare, cast, divide, copy)
                self.mask_arena.tensor.bernoulli_(self.mask_keep).div_(self.mask_keep)
        if refresh:
            self.refresh_weights()
        if self.update_running and self._bn_buffers:
            # the kernels update the running statistics through raw pointers: bump the tensors' version counters so that
            # autograd's saved-tensor checks and version-keyed caches (mds.predict) see the write
            torch.autograd.graph.increment_version(self._bn_buffers)

    _hip = None

    def _memset(self, t):
        """zero a uint8 arena on the current stream: hipMemsetAsync (one runtime fill, also a node of a captured hipGraph)"""
        if t.device.type != "cuda" or os.environ.get("MDS_MEMSET", "hip") != "hip":
            t.zero_()
            return
        if Plan._hip is None:
            import ctypes
            Plan._hip = ctypes.CDLL(_hip_path())
            Plan._hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
        rc = Plan._hip.hipMemsetAsync(t.data_ptr(), 0, t.numel() * t.element_size(), self._stream())
        if rc != 0:
            raise RuntimeError(f"hipMemsetAsync failed: {rc}")

    def begin_backward(self):
        self._memset(self._zero_bwd)

    def stale(self):
        cur = tuple(p.data_ptr() for p, *_ in self.pack_jobs)
        if self.eval_bn:
            cur += tuple(t.data_ptr() for e in self.eval_bn for t in (e.mod.weight, e.mod.bias, e.mod.running_mean, e.mod.running_var))
        return self.param_ptrs != cur
