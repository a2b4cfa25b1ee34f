"""bench.py — frame-windows/sec (fwd+bwd) of the MultiDimStacker hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

Workload (BASELINE.json configs[1], `sampling_weights_001`): per GPU a batch of 4 windows of
15x736x1280 synthetic frames (uniform [0,1) fp32, seed 1234+rank), module kwargs of
configs/ball_action/sampling_weights_001.py:30-45 (drop_rate = drop_path_rate = 0.2,
pretrained=False, random init under seed 0), bf16 autocast, focal loss (alpha -1, gamma 1.2),
backward, gradient all-reduce (N>1, RCCL) and a fused AdamW step.  A step is one pass of the hot
path over one batch; inputs are resident in HBM before the timed region.

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  roofline     : dominant kernel, algorithmic bytes (or flops) per launch / measured launch time
  cpu_baseline : the oracle (kind "port") timed on this box's host cores, bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "ball-action-spotting_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_PEAK_TFLOPS = 2500.0    # bf16 dense
FLOP_PER_WINDOW = 424.9e9    # fwd 141.6 GFLOP x 3
BYTES_PER_WINDOW_BLOCK = 1.498e9   # block-granular compulsory bytes, fwd+bwd

CONFIG = dict(model_name="tf_efficientnetv2_b0.in1k", num_classes=2, num_frames=15, stack_size=3,
              index_2d_features=4, pretrained=False, num_3d_blocks=4, num_3d_features=192,
              expansion_3d_ratio=3, se_reduce_3d_ratio=24, num_3d_stack_proj=256,
              drop_rate=0.2, drop_path_rate=0.2, act_layer="silu")


def focal_loss(logits, target, gamma=1.2):
    """sigmoid focal loss, alpha=-1, mean (reference src/losses.py:34-50) in fp32."""
    x = logits.float()
    p = torch.sigmoid(x)
    ce = torch.nn.functional.binary_cross_entropy_with_logits(x, target, reduction="none")
    p_t = p * target + (1 - p) * (1 - target)
    return (ce * (1 - p_t) ** gamma).mean()


def cpu_baseline(max_seconds=20.0):
    """The oracle (CPU restatement pinned to the reference) on the host cores, fp32 eager fwd+bwd of ONE
    full 15x736x1280 window, repeated until ~20 s of CPU work are spent (best pass reported).  If a single
    full-size pass exceeds the budget the sample falls back to a quarter of the pixels (15x368x640, same
    network and frame count) and the rate is scaled by 1/4 — the `sample` string says which was used."""
    from oracle import multidim_stacker_ref as orc
    cores = os.cpu_count() or 1
    threads = min(cores, 32)       # eager convolutions stop scaling (and oversubscribe) well before 256 threads
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    kw = dict(orc.BASIC_CONFIG_KWARGS, drop_rate=0.0, drop_path_rate=0.0)
    m = orc.MultiDimStacker(**kw).train()
    tgt = torch.tensor([[1.0, 0.0]])

    def run(h, w, budget):
        x = torch.rand(1, 15, h, w, generator=torch.Generator().manual_seed(1234))
        times, t_start = [], time.time()
        while True:
            t0 = time.time()
            m.zero_grad(set_to_none=True)
            orc.sigmoid_focal_loss(m(x), tgt, alpha=-1.0, gamma=1.2).backward()
            times.append(time.time() - t0)
            if time.time() - t_start > budget or len(times) >= 12:
                return times

    frac, shape = 1, "15x736x1280"
    times = run(736, 1280, max_seconds)
    if len(times) == 1 and times[0] > max_seconds:          # host too slow for full windows: bounded sample
        frac, shape = 4, "15x368x640"
        times = run(368, 640, max_seconds)
    timed = times[1:] if len(times) > 1 else times          # the first pass warms the allocator up
    sec = min(timed)
    scaled = "" if frac == 1 else f", scaled x1/{frac} to 15x736x1280 windows"
    return {"value": round(1.0 / (sec * frac), 5), "unit": "frame-windows/s", "cores": threads, "kind": "port",
            "sample": f"oracle fp32 eager fwd+bwd of 1 window of {shape} (batch 1), best of {len(timed)} timed passes "
                      f"({sum(times):.1f} s of CPU work in total){scaled}; host has {cores} logical cores, {threads} threads used",
            "sec_per_sample": round(sec, 3)}


def pmc_traffic(kernel_key, dtype):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/r*_pmc_hbm.json: FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs of this
    very command, KiB units, FETCH_SIZE x2 on gfx950 — MI355X_MICROARCH.md §HBM).  PMC collection
    cannot run inside the timed process, so the newest committed summary is reported; None if absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm.json")))
    if not files or dtype != "bf16":
        return None
    import re
    ks = json.load(open(files[-1]))["kernels"]
    want = kernel_key.split(".")[0] + "_kernel"

    def family(name):   # "void dw2_bwd_kernel<unsigned short, 4>(...)" -> "dw_bwd_kernel" (variants share a key)
        fn = name.replace("void ", "").split("<")[0].split("(")[0]
        fn = re.sub(r"\d", "", fn)
        for v in ("_tr_kernel", "_p_kernel", "_tiled_kernel"):
            fn = fn.replace(v, "_kernel")
        return fn

    tot = calls = 0.0
    for name, v in ks.items():
        if family(name) == want and ("unsigned short" in name or "<" not in name or "_tr_" in name or "_tiled" in name):
            tot += v["hbm_bytes_per_launch"] * v["calls"]
            calls += v["calls"]
    return int(tot / calls) if calls else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--height", type=int, default=736)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--profile-steps", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    import mds
    from mds import parallel
    torch.manual_seed(0)
    model = mds.MultiDimStacker(**CONFIG).to(dev).train()
    if world > 1:
        parallel.data_parallel(model)
    opt = torch.optim.AdamW(model.parameters(), lr=3e-4, fused=True)
    B = args.batch
    x = torch.rand(B, 15, args.height, args.width, device=dev, generator=torch.Generator(dev).manual_seed(1234 + rank))
    target = torch.randint(0, 2, (B, 2), device=dev, generator=torch.Generator(dev).manual_seed(4321)).float()
    use_amp = args.dtype == "bf16"

    def step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=use_amp):
            loss = focal_loss(model(x), target)
        loss.backward()
        opt.step()
        return loss

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    ms_per_step = elapsed / args.steps * 1e3
    wps = B * world * args.steps / elapsed
    assert torch.isfinite(loss).item(), "non-finite loss"

    # ---- per-kernel pass: HIP event pairs around every launch (on the launch stream)
    roofline, breakdown, top_launches = None, None, None
    if rank == 0 and args.profile_steps > 0:
        plan = next(p for pool in model._cache.plans.values() for p in pool if p.kind == "full" and p.need_grad)
        plan.profile = []
        for _ in range(args.profile_steps):
            step()
        torch.cuda.synchronize()
        agg = {}
        for name, seg, e0, e1, (nbytes, flops) in plan.profile:
            key = name + (".bwd" if seg.startswith("b") and name in ("pw_fwd", "conv_fwd") else "")
            a = agg.setdefault(key, [0.0, 0, 0.0, 0.0])
            a[0] += e0.elapsed_time(e1); a[1] += 1; a[2] += nbytes; a[3] += flops
        top = sorted(plan.profile, key=lambda r: -r[2].elapsed_time(r[3]))[:30]
        top_launches = [{"kernel": r[0], "seg": r[1], "us": round(r[2].elapsed_time(r[3]) * 1e3, 1),
                         "GBps": round(r[4][0] / max(r[2].elapsed_time(r[3]), 1e-6) / 1e6, 1),
                         "TFLOPs": round(r[4][1] / max(r[2].elapsed_time(r[3]), 1e-6) / 1e9, 1),
                         "MB": round(r[4][0] / 1e6, 1)} for r in top]
        plan.profile = None
        tot = sum(a[0] for a in agg.values())
        breakdown = {k: {"ms_per_step": round(a[0] / args.profile_steps, 3), "launches_per_step": a[1] // args.profile_steps,
                         "GBps": round(a[2] / a[0] / 1e6, 1) if a[0] else 0, "TFLOPs": round(a[3] / a[0] / 1e9, 2) if a[0] else 0}
                     for k, a in sorted(agg.items(), key=lambda kv: -kv[1][0])}
        dom, a = max(agg.items(), key=lambda kv: kv[1][0])
        avg_ms, avg_bytes, avg_flops = a[0] / a[1], a[2] / a[1], a[3] / a[1]
        gbs, tfl = avg_bytes / avg_ms / 1e6, avg_flops / avg_ms / 1e9
        hbm_bound = (avg_bytes / (HBM_PEAK_GBS * 1e9)) >= (avg_flops / (MFMA_PEAK_TFLOPS * 1e12))
        traffic = pmc_traffic(dom, args.dtype)
        roofline = {"kernel": dom, "bound": "hbm" if hbm_bound else "mfma",
                    "achieved": round(gbs if hbm_bound else tfl, 2), "peak": HBM_PEAK_GBS if hbm_bound else MFMA_PEAK_TFLOPS,
                    "unit": "GB/s" if hbm_bound else "TFLOP/s",
                    "frac": round((gbs / HBM_PEAK_GBS) if hbm_bound else (tfl / MFMA_PEAK_TFLOPS), 4), "traffic": traffic,
                    "avg_launch_us": round(avg_ms * 1e3, 2), "launches_per_step": a[1] // args.profile_steps,
                    "alg_bytes_per_launch": int(avg_bytes), "alg_flops_per_launch": int(avg_flops),
                    "share_of_kernel_time": round(a[0] / tot, 3),
                    "whole_path": {"mfma_frac": round(FLOP_PER_WINDOW * wps / world / (MFMA_PEAK_TFLOPS * 1e12), 4),
                                   "hbm_alg_frac_block_granular": round(BYTES_PER_WINDOW_BLOCK * wps / world / (HBM_PEAK_GBS * 1e9), 4)}}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()

    if rank == 0:
        out = {"metric": "frame-windows/sec (fwd+bwd) at 15x736x1280, batch 4", "value": round(wps, 3),
               "unit": "frame-windows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": args.dtype, "data": "synthetic",
               "config": {"workload": f"sampling_weights_001: {B}x15x{args.height}x{args.width} windows/GPU, "
                                      "fwd + focal loss + bwd + grad all-reduce + AdamW",
                          "global_batch": B * world, "parallelism": f"dp{world}", "drop_rate": 0.2, "drop_path_rate": 0.2},
               "roofline": roofline, "cpu_baseline": cpu, "kernel_breakdown": breakdown,
               "top_launches": top_launches if rank == 0 and args.profile_steps > 0 else None,
               "loss": round(float(loss.detach()), 5), "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 2)}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
