"""Developer tool (runs on the GPU box): what bounds the training step at HEAD.

  python tools/critical_path.py [--steps 12] [--only VARIANT]        -> one table on stdout

Config 2's step (bench.py's: bf16 autocast, focal loss, backward, AdamW) is timed untraced (wall clock around K steps between
synchronizes) in these variants, each made by DELETING launches from the plan's bound schedule (results become garbage, the
remaining launches keep their shapes, buffers and streams):

  base                the product step
  no_side             every weight gradient (pw_wgrad, conv_wgrad, se_fc_bwd_params) deleted: the dependent chain alone
  no_<side family>    one weight-gradient family deleted
  one_stream          MDS_SIDE_STREAM=0: the weight gradients on the chain's stream, nothing overlaps
  fwd_only            forward + loss only (no backward, no optimizer);   bwd = base - fwd_only
  chain_bwd_only      backward chain without the side ops, forward subtracted
  drop_<family>       one CHAIN family deleted (forward and backward uses), with and without the side stream running

and, with HIP event pairs around segments of the untraced step, forward / backward / optimizer wall times.  For every family
the table gives  d_step = base - variant  and, from the in-step kernel time of the family (rocprofv3 csv given with --stats, or
the column is left empty),  "1 us saved here = x us of step" = d_step / in-step kernel time.
"""
import argparse
import csv
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ball-action-spotting_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

import bench  # noqa: E402
import mds  # noqa: E402
from mds import train as mtrain  # noqa: E402

CHAIN_FAMILIES = ["bn_bwd_apply", "pw_fwd", "conv_fwd", "dw_bwd", "dw_fwd", "bn_bwd_reduce", "se_bwd_reduce", "bn_res", "se_pool",
                  "bn_finalize", "bn_bwd_finalize", "se_fc_fwd", "se_fc_bwd_data", "stem_fwd", "stem_wgrad"]
SIDE_FAMILIES = ["pw_wgrad", "conv_wgrad", "se_fc_bwd_params", "se_fc_bwd_params_table"]


def family_us(path):
    """in-step kernel time per launch-family per step from a rocprofv3 kernel_stats.csv of bench.py (10 traced steps + 3 warm-up)"""
    fam = {}
    if not path or not os.path.exists(path):
        return fam
    pats = [("pw_wgrad", r"pw_wgrad"), ("conv_wgrad", r"conv_wgrad|c3w[p2]?_kernel"), ("se_fc_bwd_params_table", r"se_bwd_b_table_kernel"), ("se_fc_bwd_params", r"se_bwd_b_kernel"), ("se_fc_bwd_data", r"se_bwd_a_kernel"),
            ("bn_bwd_apply", r"bn_bwd_apply"), ("bn_bwd_reduce", r"bn_bwd_reduce"), ("bn_bwd_finalize", r"bn_bwd_finalize"), ("bn_finalize", r"bn_finalize"),
            ("pw_fwd", r"pw_fwd|pwk"), ("conv_fwd", r"conv_fwd|c3[st]?_kernel"), ("dw_bwd", r"dw\w*_bwd"), ("dw_fwd", r"dw\w*_fwd"),
            ("se_bwd_reduce", r"se_bwd_reduce"), ("bn_res", r"bn_res"), ("se_pool", r"se_pool"), ("se_fc_fwd", r"se_fc_fwd|se_fwd"),
            ("stem_fwd", r"stem_fwd"), ("stem_wgrad", r"stem_wgrad")]
    for r in csv.DictReader(open(path)):
        name, tot, calls = r["Name"], float(r["TotalDurationNs"]), int(r["Calls"])
        if "adamw_kernel" in name:
            fam["_steps"] = calls          # one optimizer launch per traced step
        for f, pat in pats:
            if re.search(pat, name):
                fam[f] = fam.get(f, 0.0) + tot / 1e3
                break
    return fam


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--stats", default="")
    ap.add_argument("--stats-steps", type=int, default=0, help="steps the csv covers (default: the number of adamw_kernel calls in it)")
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = mds.MultiDimStacker(**bench.CONFIG).to(dev).train()
    opt = mtrain.FusedAdamW([p for p in model.parameters() if p.requires_grad], lr=3e-4)
    loss_fn = mtrain.FocalLoss(alpha=-1.0, gamma=1.2)
    x = torch.rand(4, 15, 736, 1280, device=dev, generator=torch.Generator(dev).manual_seed(1234))
    target = torch.randint(0, 2, (4, 2), device=dev, generator=torch.Generator(dev).manual_seed(4321)).float()

    def step(backward=True):
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = loss_fn(model(x), target)
        if backward:
            loss.backward()
            opt.step()

    def timed(backward=True, n=None):
        n = n or args.steps
        for _ in range(args.warmup):
            step(backward)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step(backward)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    step()
    torch.cuda.synchronize()
    plan = next(p for pool in model._cache.plans.values() for p in pool if p.kind == "full" and p.need_grad)
    orig = {seg: list(ops) for seg, ops in plan.bound.items()}

    def with_deleted(names, fn):
        """run fn() with every launch whose family is in `names` deleted from the schedule"""
        for seg, ops in orig.items():
            plan.bound[seg] = [o for o in ops if o[0].split("@")[0] not in names]
        plan._side_flags = {}
        if getattr(plan, "_ext", None) is not None:
            plan._ext.ev = {}                 # stop events are keyed by (segment, position)
        try:
            return fn()
        finally:
            for seg, ops in orig.items():
                plan.bound[seg] = list(ops)
            plan._side_flags = {}
            if getattr(plan, "_ext", None) is not None:
                plan._ext.ev = {}

    counts = {}
    for seg, ops in orig.items():
        for o in ops:
            counts[o[0].split("@")[0]] = counts.get(o[0].split("@")[0], 0) + 1
    fam_us = family_us(args.stats)
    args.stats_steps = args.stats_steps or int(fam_us.pop("_steps", 10))
    rows = []

    def record(tag, ms, fams=()):
        k_us = sum(fam_us.get(f, 0.0) for f in fams) / args.stats_steps if fams and fam_us else None
        rows.append((tag, ms, k_us, sum(counts.get(f, 0) for f in fams)))
        print(f"  {tag:28s} {ms:8.3f} ms", flush=True)

    base = timed()
    record("base", base)
    base2 = None
    if not args.only or args.only == "side":
        record("no_side", with_deleted(set(SIDE_FAMILIES), timed), SIDE_FAMILIES)
        for f in SIDE_FAMILIES:
            record(f"no_{f}", with_deleted({f}, timed), [f])
        os.environ["MDS_SIDE_STREAM"] = "0"
        record("one_stream", timed())
        record("one_stream_no_side", with_deleted(set(SIDE_FAMILIES), timed), SIDE_FAMILIES)
        os.environ["MDS_SIDE_STREAM"] = "1"
        fwd = timed(backward=False)
        record("fwd_only", fwd)
        record("chain_bwd_only(no_side - fwd)", rows[1][1] - fwd)
    if not args.only or args.only == "chain":
        base2 = timed()
        record("base (again)", base2)
        for f in CHAIN_FAMILIES:
            if counts.get(f, 0):
                record(f"drop_{f}", with_deleted({f}, timed), [f])
        for f in CHAIN_FAMILIES[:8]:
            if counts.get(f, 0):
                record(f"drop_{f}+no_side", with_deleted({f} | set(SIDE_FAMILIES), timed), [f])
    # forward / backward / optimizer wall split of the untraced product step (three event pairs per step)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(args.steps)]
    for k in range(args.steps):
        e = ev[k]
        opt.zero_grad(set_to_none=True)
        e[0].record()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = loss_fn(model(x), target)
        e[1].record()
        loss.backward()
        e[2].record()
        opt.step()
        e[3].record()
    torch.cuda.synchronize()
    seg_ms = [sum(e[j].elapsed_time(e[j + 1]) for e in ev[2:]) / (args.steps - 2) for j in range(3)]
    print("\n# what bounds the step (config 2, bf16, batch 4) - untraced wall clock per step, launches deleted from the schedule")
    print(f"# forward+loss {seg_ms[0]:.3f} ms | backward (both streams joined) {seg_ms[1]:.3f} ms | optimizer {seg_ms[2]:.3f} ms (event pairs on the caller's stream)")
    print(f"# {'variant':34s} {'ms/step':>9s} {'d_step us':>10s} {'launches':>9s} {'in-step kernel us':>18s} {'1 us saved = x us of step':>26s}")
    ns = next((r[1] for r in rows if r[0] == "no_side"), None)
    for tag, ms, k_us, nl in rows:
        ref = base2 if (base2 is not None and tag.startswith("drop_") and not tag.endswith("+no_side")) else base
        if tag.endswith("+no_side") and ns is not None:
            ref = ns
        d = (ref - ms) * 1e3
        plain = tag.startswith("base") or tag in ("fwd_only", "one_stream") or tag.startswith("chain_bwd_only")
        ratio = f"{d / k_us:26.2f}" if (k_us and not plain) else " " * 26
        kk = f"{k_us:18.1f}" if k_us else " " * 18
        dd = " " * 10 if plain else f"{d:10.1f}"
        print(f"  {tag:34s} {ms:9.3f} {dd} {nl:9d} {kk} {ratio}")


if __name__ == "__main__":
    main()
