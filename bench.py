"""bench.py — frame-windows/sec (fwd+bwd) of the MultiDimStacker hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
      N > 1 without a torchrun environment: re-executes itself under
      `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` (one rank per GPU, RCCL).

Workloads (--config):
  train    BASELINE.json configs[1] (`sampling_weights_001`): per GPU a batch of 4 windows of 15x736x1280 synthetic
           frames (uniform [0,1) fp32, seed 1234+rank), module kwargs of configs/ball_action/sampling_weights_001.py:30-45
           (drop_rate = drop_path_rate = 0.2, pretrained=False, random init under seed 0), bf16 autocast, focal loss
           (alpha -1, gamma 1.2), backward, gradient all-reduce (N>1, RCCL), AdamW step.  THE headline metric.
  long004  BASELINE.json configs[3] (`ball_finetune_long_004.py:8,67`): 4 x 33 x 736 x 1280 per GPU, 2D encoder frozen
           (forward only, BatchNorm in train mode), temporal tail forward+backward.
  predict  BASELINE.json configs[4] (src/predictors.py:50-75): sliding-window inference, one new 736x1280 frame per
           step (uint8 720x1280 ingest -> pad -> /255), frames/s, with and without horizontal-flip TTA.
A step is one pass of the hot path over one batch; inputs are resident in HBM before the timed region.

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  roofline     : the dominant HIP kernel family IN THE STEP (rocprofv3 --kernel-trace --stats child run of this very script,
                 both HIP streams running; fwd and bwd uses of one kernel folded together): algorithmic bytes (or flops)
                 per launch / its average in-step launch duration; `isolated` keeps the per-launch HIP-event numbers (each
                 launch timed alone); PMC traffic from further child runs; the whole-path fraction of SURVEY.md §8(d)
  other_configs: BASELINE.json configs[3] (long004, its own SGD-Nesterov recipe) and configs[4] (predict: reference API
                 frame by frame and predict_batch, with its cpu_baseline), each a child run of this script
  cpu_baseline : the oracle (kind "port") timed on this box's host cores, bounded sample.
"""
import argparse
import csv
import glob
import json
import contextlib
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "ball-action-spotting_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_PEAK_TFLOPS = 2500.0    # bf16 dense
# SURVEY.md §8(d) / BASELINE.md §2: algorithmic work per window (block-granular compulsory bytes)
WORK = {"train": dict(flop=424.9e9, bytes=1.498e9), "long004": dict(flop=353.4e9, bytes=1.204e9)}

CONFIG = dict(model_name="tf_efficientnetv2_b0.in1k", num_classes=2, num_frames=15, stack_size=3,
              index_2d_features=4, pretrained=False, num_3d_blocks=4, num_3d_features=192,
              expansion_3d_ratio=3, se_reduce_3d_ratio=24, num_3d_stack_proj=256,
              drop_rate=0.2, drop_path_rate=0.2, act_layer="silu")


def focal_loss(logits, target, gamma=1.2, alpha=-1.0):
    """sigmoid focal loss, mean (reference src/losses.py:34-50) in fp32."""
    import torch
    x = logits.float()
    p = torch.sigmoid(x)
    ce = torch.nn.functional.binary_cross_entropy_with_logits(x, target, reduction="none")
    p_t = p * target + (1 - p) * (1 - target)
    loss = ce * (1 - p_t) ** gamma
    if alpha >= 0:
        loss = (alpha * target + (1 - alpha) * (1 - target)) * loss
    return loss.mean()


def cpu_baseline(max_seconds=20.0, frames=15, frozen=False):
    """The oracle (CPU restatement pinned to the reference) on the host cores, fp32 eager fwd+bwd of ONE
    full 15x736x1280 window (config 4: 33 frames, 2D encoder frozen, BatchNorm in train mode), repeated until ~20 s of
    CPU work are spent (best pass reported).  If a single
    full-size pass exceeds the budget the sample falls back to a quarter of the pixels (15x368x640, same
    network and frame count) and the rate is scaled by 1/4 — the `sample` string says which was used.
    A bf16-autocast pass of the same window is timed beside it (SURVEY §8d asks for both)."""
    import torch
    from oracle import multidim_stacker_ref as orc
    cores = os.cpu_count() or 1
    # eager convolutions stop scaling (and oversubscribe) well before 256 threads.  Measured on the GPU box's host (256 logical cores,
    # profiles/r05_cpu_threads.txt): one fp32 fwd+bwd window takes 2.4 s at 32 threads, 5.1 s at 64, 10.8 s at 128 and 237 s at 256
    threads = min(cores, 32)
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    kw = dict(orc.BASIC_CONFIG_KWARGS, drop_rate=0.0, drop_path_rate=0.0, num_frames=frames)
    m = orc.MultiDimStacker(**kw).train()
    if frozen:
        for p_ in m.conv2d_encoder.parameters():
            p_.requires_grad_(False)
    tgt = torch.tensor([[1.0, 0.0]])

    def run(h, w, budget, amp=False, max_n=12):
        x = torch.rand(1, frames, h, w, generator=torch.Generator().manual_seed(1234))
        times, t_start = [], time.time()
        while True:
            t0 = time.time()
            m.zero_grad(set_to_none=True)
            with torch.autocast("cpu", dtype=torch.bfloat16, enabled=amp):
                out = m(x)
            orc.sigmoid_focal_loss(out.float(), tgt, alpha=-1.0, gamma=1.2).backward()
            times.append(time.time() - t0)
            if time.time() - t_start > budget or len(times) >= max_n:
                return times

    frac, shape = 1, f"{frames}x736x1280"
    times = run(736, 1280, max_seconds)
    if len(times) == 1 and times[0] > max_seconds:          # host too slow for full windows: bounded sample
        frac, shape = 4, f"{frames}x368x640"
        times = run(368, 640, max_seconds)
    timed = times[1:] if len(times) > 1 else times          # the first pass warms the allocator up
    sec = min(timed)
    scaled = "" if frac == 1 else f", scaled x1/{frac} to {frames}x736x1280 windows"
    out = {"value": round(1.0 / (sec * frac), 5), "unit": "frame-windows/s", "cores": threads, "kind": "port",
           "host_logical_cores": cores,
           "sample": f"oracle fp32 eager {'encoder fwd (frozen, train-mode BN) + tail fwd+bwd' if frozen else 'fwd+bwd'} of 1 window of {shape} (batch 1), best of {len(timed)} timed passes "
                     f"({sum(times):.1f} s of CPU work in total){scaled}; {threads} threads of {cores} logical cores used",
           "sec_per_sample": round(sec, 3)}
    if frozen:
        return out
    try:
        tb = run(736, 1280, 6.0, amp=True, max_n=4)        # the full window (round 5: it is the faster of the two on this host)
        out["bf16_autocast"] = {"value": round(1.0 / min(tb), 5), "unit": "frame-windows/s",
                                "sample": f"same oracle under torch.autocast('cpu', bfloat16), 1 window of {shape}, "
                                          f"best of {len(tb)} passes ({sum(tb):.1f} s)"}
    except Exception as e:  # CPU bf16 convolutions may be unsupported on an old host
        out["bf16_autocast"] = {"error": str(e)[:120]}
    return out


# ----------------------------------------------------------------------------------------------- PMC child runs
def kernel_family(name):
    """'void dw2_bwd_kernel<unsigned short, 4>(...)' -> 'dw_bwd' ; variants of one entry point share a key"""
    fn = name.replace("void ", "").split("<")[0].split("(")[0]
    if fn in ("c3_kernel", "c3t_kernel", "c3s_kernel"):      # k_c3.hip serves mds_conv_fwd launches and (its one-tap form: template argument ONE, the 11th) a few of mds_pw_fwd's
        targs = [t.strip() for t in name.split("<", 1)[1].split(">")[0].split(",")] if "<" in name else []
        return "pw_fwd" if fn == "c3_kernel" and len(targs) > 10 and targs[10] == "true" else "conv_fwd"
    if fn in ("c3w_kernel", "c3wp_kernel", "c3w2_kernel"):    # ... and mds_conv_wgrad's launches
        return "conv_wgrad"
    if fn == "se_bwd_b_table_kernel":          # the table form of se_bwd_b_kernel (one launch per gradient bucket)
        return "se_bwd_b"
    fn = re.sub(r"\d", "", fn)
    for v in ("_tr_kernel", "_p_kernel", "_tiled_kernel", "_wres_kernel", "_q_kernel", "_kernel"):
        if fn.endswith(v):
            fn = fn[: -len(v)]
            break
    return fn.replace("dws_", "dw_")


def pmc_child(counters, extra_args, timeout=120, steps=2):
    """Run this script under `rocprofv3 --pmc <counters>` (counter collection only — never combined with trace
    domains) for 2 steps and return {kernel family: {counter: mean per launch, 'launches': n, 'us': mean duration}}."""
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None
    tmp = tempfile.mkdtemp(prefix="mds_pmc_", dir="/tmp")
    cmd = [exe, "--pmc", *counters, "-d", tmp, "-o", "pmc", "--output-format", "csv", "--", sys.executable,
           os.path.abspath(__file__), "--steps", str(steps), "--warmup", "1", "--profile-steps", "0", "--no-cpu-baseline", "--no-pmc",
           "--no-other-configs", *extra_args]      # (the config 4 / 5 child runs inside a counter pass took it past its timeout)
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    try:
        subprocess.run(cmd, cwd="/tmp", env=env, timeout=timeout, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        files = glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            return None
        agg, seen = {}, set()
        for r in csv.DictReader(open(files[0])):
            fam = kernel_family(r["Kernel_Name"])
            e = agg.setdefault(fam, {"launches": set(), "dur": 0.0})
            e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            d = r["Dispatch_Id"]
            if d not in seen:
                seen.add(d)
                e["launches"].add(d)
                e["dur"] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-3
        out = {}
        for fam, e in agg.items():
            n = max(len(e["launches"]), 1)
            out[fam] = {k: v / n for k, v in e.items() if k not in ("launches", "dur")}
            out[fam]["launches"] = n
            out[fam]["us"] = e["dur"] / n
        return out
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def trace_child(extra_args, steps=8, timeout=300):
    """Run this script under `rocprofv3 --kernel-trace --stats` (both HIP streams running, the real step) and return
    {kernel family: {"calls": n, "total_us": t, "avg_us": t / n}} from its kernel_stats.csv - the IN-STEP launch durations."""
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None
    tmp = tempfile.mkdtemp(prefix="mds_kt_", dir="/tmp")
    cmd = [exe, "--kernel-trace", "--stats", "-d", tmp, "-o", "kt", "--output-format", "csv", "--", sys.executable,
           os.path.abspath(__file__), "--steps", str(steps), "--warmup", "2", "--profile-steps", "0", "--no-cpu-baseline", "--no-pmc",
           "--no-other-configs", *extra_args]
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    try:
        subprocess.run(cmd, cwd="/tmp", env=env, timeout=timeout, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        files = glob.glob(os.path.join(tmp, "**", "*kernel_stats.csv"), recursive=True)
        if not files:
            return None
        if os.environ.get("MDS_KEEP_TRACE_STATS"):      # developer: keep the csv.  The variable is a PREFIX (relative: from the repo
            # root); the traced config names the file, so the child runs of configs 4 / 5 cannot overwrite the training step's
            cfg = extra_args[extra_args.index("--config") + 1] if "--config" in extra_args else "train"
            tag = {"train": "bench", "predict": "predict_fbf"}.get(cfg, cfg)
            shutil.copy(files[0], os.path.join(ROOT, f'{os.environ["MDS_KEEP_TRACE_STATS"]}_{tag}_kernel_stats.csv'))
        out = {}
        for r in csv.DictReader(open(files[0])):
            fam = kernel_family(r["Name"])
            e = out.setdefault(fam, {"calls": 0, "total_us": 0.0})
            e["calls"] += int(r["Calls"]); e["total_us"] += float(r["TotalDurationNs"]) * 1e-3
        for e in out.values():
            e["avg_us"] = e["total_us"] / max(e["calls"], 1)
        out["_steps"] = steps + 2
        return out
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def other_config(name, extra, timeout=600):
    """One more BASELINE.json config measured by a child run of this script (fresh process: the 12 GB plan of the main
    workload is gone); returns its JSON line as a dict with the bulky per-kernel fields dropped."""
    cmd = [sys.executable, os.path.abspath(__file__), "--config", name, "--no-pmc", "--profile-steps", "0", "--no-other-configs", *extra]
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    try:
        r = subprocess.run(cmd, env=env, timeout=timeout, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        d = json.loads(line)
        for k in ("kernel_breakdown", "top_launches"):
            d.pop(k, None)
        return d
    except Exception as e:
        return {"error": f"{type(e).__name__}: {str(e)[:200]}"}


def pmc_for(dom, extra_args):
    """HBM traffic per launch of kernel family `dom` (FETCH_SIZE and WRITE_SIZE in separate passes, KiB units,
    FETCH_SIZE x2 on gfx950 — MI355X_MICROARCH.md §HBM) and its MFMA-busy / wave-cycle split."""
    res = {"traffic": None}
    f = pmc_child(["FETCH_SIZE"], extra_args) or pmc_child(["FETCH_SIZE"], extra_args)      # (one retry: a counter pass on a fresh box
    w = pmc_child(["WRITE_SIZE"], extra_args) or pmc_child(["WRITE_SIZE"], extra_args)      #  occasionally comes back without its csv)
    if f and w and dom in f and dom in w:
        res["traffic"] = int(f[dom]["FETCH_SIZE"] * 1024 * 2 + w[dom]["WRITE_SIZE"] * 1024)
        nsteps = 3.0      # the child runs 1 warm-up + 2 timed steps
        res["step_traffic_GB"] = round((sum(f[k]["FETCH_SIZE"] * 2048 * f[k]["launches"] for k in f) +
                                        sum(w[k]["WRITE_SIZE"] * 1024 * w[k]["launches"] for k in w)) / nsteps / 1e9, 2)
    s = pmc_child(["SQ_WAVE_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"], extra_args)
    if s and dom in s and s[dom].get("SQ_WAVE_CYCLES"):
        e = s[dom]
        res["sq"] = {"mfma_busy_frac_at_2.4GHz": round(e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024 * e["us"] * 1e-6 * 2.4e9), 4),
                     "wave_parked": round(e.get("SQ_WAIT_ANY", 0.0) / e["SQ_WAVE_CYCLES"], 3),
                     "issue_stall": round(e.get("SQ_WAIT_INST_ANY", 0.0) / e["SQ_WAVE_CYCLES"], 3),
                     "issuing": round(e.get("SQ_ACTIVE_INST_ANY", 0.0) / e["SQ_WAVE_CYCLES"], 3),
                     "profiled_us": round(e["us"], 2)}
    return res


# ----------------------------------------------------------------------------------------------- config 5: sliding-window inference
def bench_predict(args, dev, rank, world):
    """BASELINE.json configs[4] (src/predictors.py:50-75): a stream of raw 720x1280 uint8 frames, one new frame per step
    through mds.predict.StreamPredictor (fused ingest, device feature store, eval BN table, hipGraph replay); replicas
    only across GPUs (each rank its own stream, no collective).  value = frames/s without TTA in fp32 (the reference's
    predictor runs outside autocast); the TTA and bf16 rates and the module-call path of round 1 are reported beside it."""
    import torch
    import torch.distributed as dist
    import mds
    from mds.predict import StreamPredictor
    torch.manual_seed(0)
    model = mds.MultiDimStacker(**dict(CONFIG, drop_rate=0.0, drop_path_rate=0.0)).to(dev)
    # realistic running statistics: one training-mode pass with momentum 1 (random-init running stats blow eval mode up)
    for bn in model.modules():
        if isinstance(bn, torch.nn.modules.batchnorm._BatchNorm):
            bn.momentum = 1.0
    model.train()
    with torch.no_grad():
        model(torch.rand(1, 15, 736, 1280, device=dev))
    model.eval()
    model.clear_plans()
    pool = torch.randint(0, 256, (64, 720, 1280), dtype=torch.uint8, device=dev, generator=torch.Generator(dev).manual_seed(99 + rank))
    K, Wm = args.steps, max(args.warmup, 40)          # the first 28 frames only fill the window

    CH = args.chunk

    last = {}

    def run(tta, cdt, graphs=True, chunk=CH, pipelined=0):
        """K frames through the predictor, `chunk` consecutive frames per call (1 = the reference's frame-by-frame API);
        pipelined: through StreamPredictor.predict_stream (the same passes, encoder of step j + 1 beside the tail of step j)"""
        sp = StreamPredictor(model, frame_size=(1280, 736), tta=tta, compute_dtype=cdt, use_graphs=graphs and os.environ.get("MDS_BENCH_NO_GRAPH", "0") != "1")
        last["sp"] = sp
        idx = 0

        def feed(nfr):
            nonlocal idx
            out = None
            done = 0
            if pipelined:
                first = idx
                for out, _ in sp.predict_stream((pool[(first + j) % 64] for j in range(nfr)), first, chunk=chunk, lanes=pipelined):
                    pass
                idx += nfr
                return out
            while done < nfr:
                c = min(chunk, nfr - done)
                a = idx % 64            # the caller's frames: a VIEW of the pool (no arange / remainder / gather launches of the
                if a + c <= 64:         # bench's own in the timed region - they were 3 of the 105 launches of a frame)
                    fr = pool[a:a + c]
                else:
                    fr = torch.cat([pool[a:], pool[:a + c - 64]])
                out = sp.predict_batch(fr, idx)[-1][0]
                idx += c; done += c
            return out
        W = max(Wm, 28 + 6 * chunk, 6 * chunk * max(pipelined, 1) + 28)       # every lane past its eager passes and its hipGraph capture
        feed(-(-W // chunk) * chunk + K % chunk)                               # + one chunk of the size the K timed frames end with (its plans exist)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = feed(K)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        assert out is not None and torch.isfinite(out).all()
        return el

    def module_path(tta):      # round 1's path: forward_2d / forward_3d / forward_head module calls on padded fp32 frames
        b = 2 if tta else 1
        feats = [torch.randn(b, 1, 192, 23, 40, device=dev) for _ in range(5)]
        frames = torch.rand(b, 3, 736, 1280, device=dev)
        def one():
            with torch.no_grad():
                f = model.forward_2d(frames)
                feats.pop(0); feats.append(f)
                return model.forward_head(model.forward_3d(torch.cat(feats, dim=1)))
        for _ in range(5):
            one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            one()
        torch.cuda.synchronize()
        return 50 / (time.perf_counter() - t0)

    if args.predict_pipe_only:       # developer: the pipelined rates alone
        if rank == 0:
            out = {"fbf": round(K / run(False, None, chunk=1), 1), f"chunk{CH}": round(K / run(False, None), 1)}
            for L in (1, 2, 3, 4):
                out[f"fbf_lanes{L}"] = round(K / run(False, None, chunk=1, pipelined=L), 1)
                out[f"chunk{CH}_lanes{L}"] = round(K / run(False, None, pipelined=L), 1)
            out["fbf_tta_lanes3"] = round(K / run(True, None, chunk=1, pipelined=3), 1)
            print(json.dumps(out))
        return
    if args.predict_fbf_only:        # the kernel-trace child: only the reference's frame-by-frame API, fp32
        el = run(False, None, chunk=1)
        if rank == 0:
            print(json.dumps({"metric": "frames/sec, frame-by-frame API (trace child)", "value": round(K / el, 2), "unit": "frames/s"}))
        return
    fbf = None
    if rank == 0 and world == 1:     # the plain calls first (a fresh process, as a caller of predict() has it)
        fbf = {"note": "predict(frame, index) calls, each result consumed in order on the caller's stream (no look-ahead)",
               "fp32_frames_per_s": round(K / run(False, None, chunk=1), 1),
               "fp32_tta_frames_per_s": round(K / run(True, None, chunk=1), 1),
               "bf16_frames_per_s": round(K / run(False, "bf16", chunk=1), 1)}
    LN = args.lanes or (4 if CH <= 4 else (3 if CH <= 8 else 2))
    L1 = args.lanes or 4          # lanes of the one-frame-per-pass runs
    el = run(False, None, pipelined=LN)
    lanes_in_use = getattr(last.get("sp"), "lanes_in_use", None)      # what the lane selection gave this process (mds.predict._lane_streams)
    if world > 1:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = t.item()
    fps = K * world / el
    extra = {}
    if rank == 0 and world == 1:
        extra = {"fp32_tta_frames_per_s": round(K / run(True, None, pipelined=LN), 1), "bf16_frames_per_s": round(K / run(False, "bf16", pipelined=LN), 1),
                 "one_stream_no_lanes": {"note": f"predict_batch calls on the caller's stream, {CH} frames per call (rounds 2-4's headline path)",
                                         "fp32_frames_per_s": round(K / run(False, None), 1)},
                 "one_frame_per_pass": {"note": "the reference's pattern (one new stack through the 2D encoder, one window through the tail per frame) "
                                                f"through predict_stream, {L1} lanes",
                                        "fp32_frames_per_s": round(K / run(False, None, chunk=1, pipelined=L1), 1),
                                        "fp32_tta_frames_per_s": round(K / run(True, None, chunk=1, pipelined=L1), 1)},
                 "frame_by_frame_api": fbf,
                 "round1_module_call_path_frames_per_s": round(module_path(False), 1)}
        from mds import predict as _mp
        extra["lanes"] = {"requested": LN, "in_use": lanes_in_use, "selection": _mp.lane_log(),
                          "tta_on_is_the_reference_setting": "scripts/ball_action/predict.py:16 TTA = True: fp32_tta_frames_per_s is config 5 as the reference runs it"}
    kroof = None
    if rank == 0 and world == 1 and args.predict_kernel_trace:
        # per-kernel roofline of the reference's frame-by-frame API: algorithmic bytes / flops per launch from the plans'
        # own cost table, launch durations from a rocprofv3 --kernel-trace --stats child run of this script on that path
        fbf_fps = K / run(False, None, chunk=1)
        sp = last["sp"]
        cost = {}
        for c in sp.plans.values():
            for plan in (c["p2d"][0], c["ptail"][0]):
                for seg, ops in plan.bound.items():
                    for (name, *_), (nb, fl) in zip(ops, plan.costs[seg]):
                        e = cost.setdefault(kernel_family(name + "_kernel"), [0, 0.0, 0.0])
                        e[0] += 1; e[1] += nb; e[2] += fl
        kt = trace_child(["--config", "predict", "--predict-fbf-only"], steps=300, timeout=400)
        if kt:
            kt.pop("_steps")
            fams = {k: v for k, v in kt.items() if k in cost and cost[k][1] > 0}
            tot_us = sum(v["total_us"] for v in kt.values())
            if fams:
                dom = max(fams, key=lambda k: fams[k]["total_us"])
                n, nb, fl = cost[dom]
                avg_b, avg_f, avg_us = nb / n, fl / n, fams[dom]["avg_us"]
                gbs, tfl = avg_b / avg_us / 1e3, avg_f / avg_us / 1e6
                hbm = avg_b / (HBM_PEAK_GBS * 1e9) >= avg_f / (MFMA_PEAK_TFLOPS * 1e12)
                kroof = {"kernel": dom, "bound": "hbm" if hbm else "mfma", "achieved": round(gbs if hbm else tfl, 2),
                         "peak": HBM_PEAK_GBS if hbm else MFMA_PEAK_TFLOPS, "unit": "GB/s" if hbm else "TFLOP/s",
                         "frac": round(gbs / HBM_PEAK_GBS if hbm else tfl / MFMA_PEAK_TFLOPS, 5), "traffic": None,
                         "avg_launch_us": round(avg_us, 2), "launches_per_frame": n, "alg_bytes_per_launch": int(avg_b),
                         "alg_flops_per_launch": int(avg_f), "share_of_kernel_time": round(fams[dom]["total_us"] / tot_us, 3),
                         "path": f"frame by frame (the reference's API), fp32, {fbf_fps:.0f} frames/s; durations: rocprofv3 --kernel-trace --stats child run",
                         "us_per_frame_by_family": {k: round(v["avg_us"] * cost[k][0], 1) for k, v in sorted(fams.items(), key=lambda kv: -kv[1]["total_us"])[:8]}}
                # HBM traffic per launch of that kernel: FETCH_SIZE / WRITE_SIZE passes of the same path (separate runs, KiB units,
                # FETCH_SIZE x2 on gfx950 - MI355X_MICROARCH.md HBM section); 40 frames per pass: rocprofv3 7.2 segfaults in a 300-frame
                # counter pass and (round 5) did not come back from a 100-frame one within 15 minutes
                pa = ["--config", "predict", "--predict-fbf-only"]
                f = pmc_child(["FETCH_SIZE"], pa, timeout=120, steps=40) or pmc_child(["FETCH_SIZE"], pa, timeout=120, steps=40)
                w = pmc_child(["WRITE_SIZE"], pa, timeout=120, steps=40) or pmc_child(["WRITE_SIZE"], pa, timeout=120, steps=40)
                if f and w and dom in f and dom in w:
                    kroof["traffic"] = int(f[dom]["FETCH_SIZE"] * 2048 + w[dom]["WRITE_SIZE"] * 1024)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import multidim_stacker_ref as orc
        torch.set_num_threads(min(os.cpu_count() or 1, 32))
        ref = orc.MultiDimStacker(**dict(orc.BASIC_CONFIG_KWARGS, drop_rate=0.0, drop_path_rate=0.0)).eval()
        fr = torch.rand(1, 3, 736, 1280); feats = [torch.randn(1, 1, 192, 23, 40) for _ in range(5)]
        ts = []
        with torch.no_grad():
            for _ in range(6):
                t0 = time.time()
                f = ref.forward_2d(fr); feats.pop(0); feats.append(f)
                ref.forward_head(ref.forward_3d(torch.cat(feats, dim=1)))
                ts.append(time.time() - t0)
        cpu = {"value": round(1.0 / min(ts[1:]), 3), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
               "host_logical_cores": os.cpu_count(),
               "sample": f"oracle fp32, predictor access pattern (one new 3x736x1280 stack + tail on 5 cached stacks), best of 5 frames ({sum(ts):.1f} s)"}
    if rank == 0:
        t_frame = el / K
        out = {"metric": "frames/sec, sliding-window inference (src/predictors.py) on raw 720x1280 frames padded to 736x1280", "value": round(fps, 2),
               "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": round(t_frame * 1e3, 4), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f32 (1x1 / 3x3 products as split-bf16: three bf16 MFMAs, >= 16 significant bits)", "data": "synthetic",
               "config": {"workload": f"sliding-window predictor over a stream of raw uint8 frames, 15-frame window stride 2, no TTA, fp32, {CH} consecutive "
                                      f"frames per pass, {LN} lanes (StreamPredictor.predict_stream: offline prediction of a half, throughput-only; "
                                      "a step = one frame); one independent stream of frames per GPU",
                          "parallelism": f"replicas x{world}"},
               "roofline": {"bound": "mfma", "frac_whole_path": round(max(35.9e9 / (MFMA_PEAK_TFLOPS * 1e12), 0.12e9 / (HBM_PEAK_GBS * 1e9)) / t_frame, 5),
                            "definition": "SURVEY.md 8(d) config 5: 35.9 GFLOP, 0.12 GB per frame; launch-latency bound in practice",
                            **(kroof or {"kernel": None, "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": None})},
               "cpu_baseline": cpu, **extra}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


# ----------------------------------------------------------------------------------------------- main
def respawn(args):
    """`python bench.py --gpus N` (N > 1) outside torchrun: launch one rank per GPU ourselves."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    sys.exit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 20; 300 frames for --config predict)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="train", choices=["train", "long004", "predict"])
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--height", type=int, default=736)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--profile-steps", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 child runs that measure roofline.traffic and the in-step kernel durations")
    ap.add_argument("--trace-only", action="store_true", help="with --no-pmc: still take the in-step kernel durations (rocprofv3 --kernel-trace "
                    "child run) for the per-kernel roofline; only the counter passes are skipped (the config 4 child run)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the child runs of configs 4 and 5 (`other_configs` of the train line)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--n1-ms", type=float, default=None, help="--gpus N > 1: ms per step of an N = 1 run, for parallel.efficiency_vs_n1 "
                    "(default: measured in the same run as steps without the exchange)")
    ap.add_argument("--predict-pipe-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--lanes", type=int, default=0, help="--config predict: lanes of StreamPredictor.predict_stream (steps in flight on their own HIP streams; 0 = its own rule)")
    ap.add_argument("--chunk", type=int, default=8, help="--config predict: consecutive frames per predictor call")
    ap.add_argument("--torch-step", action="store_true", help="torch's focal loss + torch.optim.AdamW(fused=True) instead of mds.train")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="CPU work budget of the cpu_baseline leg")
    ap.add_argument("--predict-kernel-trace", action="store_true", help="--config predict: rocprofv3 kernel-trace child of the frame-by-frame path -> per-kernel roofline")
    ap.add_argument("--predict-fbf-only", action="store_true", help="--config predict: time only the reference's frame-by-frame API in fp32 (the trace child)")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 300 if args.config == "predict" else 20

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn(args)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # MDS_BENCH_SHARED_GPU=1 (developer, FUNCTIONAL test of the N > 1 code path on a one-GPU box): every rank on device 0, slices
    # over gloo (RCCL refuses two ranks on one device); the line says so and is not a measurement of N GPUs
    shared = os.environ.get("MDS_BENCH_SHARED_GPU", "0") == "1"
    if shared:
        local = 0
    assert shared or torch.cuda.device_count() >= args.gpus, f"--gpus {args.gpus} but only {torch.cuda.device_count()} visible devices (one rank per GPU)"
    assert local < torch.cuda.device_count(), f"LOCAL_RANK {local} has no device of its own"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if shared:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        assert dist.get_world_size() == args.gpus == world, (dist.get_world_size(), args.gpus, world)
        devs = [None] * world                       # every rank on its own device: N ranks sharing one GPU would not be an N-GPU line
        dist.all_gather_object(devs, (os.uname().nodename, torch.cuda.current_device()))
        assert shared or len(set(devs)) == world, f"ranks share devices: {devs}"

    if args.config == "predict":
        return bench_predict(args, dev, rank, world)

    import mds
    from mds import parallel
    torch.manual_seed(0)
    cfg = dict(CONFIG)
    T = 15
    if args.config == "long004":
        T = 33
        cfg.update(num_frames=33)
    model = mds.MultiDimStacker(**cfg).to(dev).train()
    if args.config == "long004":      # src/argus_models.py:104-110: freeze flips requires_grad only; BN stays in train mode
        for p in model.conv2d_encoder.parameters():
            p.requires_grad_(False)
    if world > 1:
        parallel.data_parallel(model)
    from mds import train as mtrain
    trainable = [p for p in model.parameters() if p.requires_grad]
    long = args.config == "long004"
    # the config's own recipe: sampling_weights_001.py:46-55 AdamW + focal(alpha -1, gamma 1.2); ball_finetune_long_004.py:46-55
    # SGD(momentum 0.9, nesterov) at lr = get_lr(1e-3, 4) + focal(alpha 0.4, gamma 1.2)
    lr = 1e-3 * 4 / 4 if long else 3e-4
    if args.torch_step:
        opt = torch.optim.SGD(trainable, lr=lr, momentum=0.9, nesterov=True) if long else torch.optim.AdamW(trainable, lr=lr, fused=True)
        loss_fn = (lambda o, t: focal_loss(o, t, alpha=0.4)) if long else focal_loss
    else:      # loss value+gradient in one launch, the optimizer over all tensors in one launch
        opt = mtrain.FusedSGD(trainable, lr=lr, momentum=0.9, nesterov=True) if long else mtrain.FusedAdamW(trainable, lr=lr)
        loss_fn = mtrain.FocalLoss(alpha=0.4 if long else -1.0, gamma=1.2)
    B = args.batch
    x = torch.rand(B, T, args.height, args.width, device=dev, generator=torch.Generator(dev).manual_seed(1234 + rank))
    target = torch.randint(0, 2, (B, 2), device=dev, generator=torch.Generator(dev).manual_seed(4321)).float()
    use_amp = args.dtype == "bf16"

    def step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=use_amp):
            loss = loss_fn(model(x), target)
        loss.backward()
        opt.step()
        return loss

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # MDS_BENCH_STREAM=1 (developer A/B): the steps run on a non-default HIP stream, as a trainer that owns its stream would
    own = torch.cuda.stream(torch.cuda.Stream(dev)) if os.environ.get("MDS_BENCH_STREAM", "0") == "1" else contextlib.nullcontext()
    with own:
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
    par_info = None
    if world > 1:       # one more step with an event pair around every slice's all-reduce: makes an N-GPU line self-explaining
        sync = getattr(model, "_grad_sync", None)
        if hasattr(sync, "timing"):
            sync.timing = True
            step()
            fence()
            sync.timing = False
            sl = sync.last_timing or []
            # the same steps WITHOUT the exchange on this very device = what an N = 1 run measures (weak scaling: per-rank work is
            # fixed), so the line carries its own efficiency and the stated model's prediction next to the measurement
            sync.paused = True
            for _ in range(2):
                step()
            fence()
            t0 = time.perf_counter()
            nloc = max(args.steps // 2, 3)
            for _ in range(nloc):
                step()
            fence()
            tl = torch.tensor([(time.perf_counter() - t0) / nloc * 1e3], device=dev, dtype=torch.float64)
            dist.all_reduce(tl, op=dist.ReduceOp.MAX)
            local_ms = args.n1_ms if args.n1_ms else tl.item()
            sync.paused = False
            model_pred = parallel.predict_step_ms(local_ms, world, [(hi - lo) * 4 for lo, hi, _ in sl])
            par_info = {"rccl_ranks": world, "backend": dist.get_backend(), "devices_in_use": 1 if shared else world,
                        "local_step_ms_no_exchange": round(tl.item(), 4), "n1_ms_given": args.n1_ms,
                        "efficiency_vs_n1": round(local_ms / ms_per_step, 4), **model_pred,
                        "allreduce_slices_in_backward_order":
                        [{"elems": hi - lo, "MB": round((hi - lo) * 4 / 1e6, 2), "us": us} for lo, hi, us in sl],
                        "allreduce_exposed_ms": sync.last_exposed_ms,
                        "note": "rank 0, communication stream; the slices overlap the rest of the backward pass; allreduce_exposed_ms = end of the "
                                "last backward kernel on the compute stream -> end of the last all-reduce (what the step waits for); model: "
                                "27.1 MB ring all-reduce over xGMI ~0.31 ms at 8 ranks, all but the last slice hidden (DESIGN 2e)"}
    assert torch.isfinite(loss).item(), "non-finite loss"

    # ---- per-kernel pass: HIP event pairs around every launch (on the launch stream).  Keyed by HIP kernel:
    #      the data-gradient launches of pw_fwd / conv_fwd are the same kernels as the forward ones.
    roofline, breakdown, top_launches = None, None, None
    full = (args.height, args.width) == (736, 1280)
    work = WORK[args.config] if full else None
    if world > 1 and getattr(model, "_grad_sync", None) is not None and hasattr(model._grad_sync, "paused"):
        model._grad_sync.paused = True      # the per-kernel pass below runs on rank 0 ONLY: its steps must not enter a collective the
                                            # other ranks never join (rank 0 would hang in the next synchronize)
    if rank == 0 and args.profile_steps > 0:
        plan = next(p for pool in model._cache.plans.values() for p in pool if p.kind == "full" and p.need_grad)
        plan.profile = []
        for _ in range(args.profile_steps):
            step()
        torch.cuda.synchronize()
        agg = {}
        for name, seg, e0, e1, (nbytes, flops) in plan.profile:
            a = agg.setdefault(name, [0.0, 0, 0.0, 0.0])
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
        roofline = {"kernel": dom, "bound": "hbm" if hbm_bound else "mfma",
                    "achieved": round(gbs if hbm_bound else tfl, 2), "peak": HBM_PEAK_GBS if hbm_bound else MFMA_PEAK_TFLOPS,
                    "unit": "GB/s" if hbm_bound else "TFLOP/s",
                    "frac": round((gbs / HBM_PEAK_GBS) if hbm_bound else (tfl / MFMA_PEAK_TFLOPS), 4), "traffic": None,
                    "avg_launch_us": round(avg_ms * 1e3, 2), "launches_per_step": a[1] // args.profile_steps,
                    "alg_bytes_per_launch": int(avg_bytes), "alg_flops_per_launch": int(avg_flops),
                    "kernel_TFLOPs": round(tfl, 1), "kernel_mfma_frac": round(tfl / MFMA_PEAK_TFLOPS, 4),
                    "share_of_kernel_time": round(a[0] / tot, 3)}
        if work:
            t_window = elapsed / args.steps / B
            roofline["frac_whole_path"] = round(max(work["flop"] / (MFMA_PEAK_TFLOPS * 1e12), work["bytes"] / (HBM_PEAK_GBS * 1e9)) / t_window, 4)
            roofline["whole_path"] = {"mfma_frac": round(work["flop"] * wps / world / (MFMA_PEAK_TFLOPS * 1e12), 4),
                                      "hbm_alg_frac_block_granular": round(work["bytes"] * wps / world / (HBM_PEAK_GBS * 1e9), 4),
                                      "definition": "SURVEY.md 8(d): max(F/2.5e15, B/8e12) / t_window"}
        if world == 1 and (not args.no_pmc or args.trace_only):
            extra = ["--config", args.config, "--batch", str(B), "--height", str(args.height), "--width", str(args.width),
                     "--dtype", args.dtype] + (["--torch-step"] if args.torch_step else [])
            # IN-STEP durations: the per-kernel pass above times each launch ALONE on one stream; inside the step the weight
            # gradients run on a second stream beside the dependent chain.  A rocprofv3 --kernel-trace --stats child run of this
            # script gives every kernel's duration in the real step; the dominant family and the roofline fraction come from
            # THAT trace (profiles/rNN_bench_kernel_stats.csv is the same command), the isolated numbers stay as `isolated`.
            kt = trace_child(extra)
            if kt:
                nst = kt.pop("_steps")
                fams = {k: v for k, v in kt.items() if k in agg}
                if fams:
                    isolated = dict(roofline)
                    dom = max(fams, key=lambda k: fams[k]["total_us"])
                    a = agg[dom]
                    n_launch = a[1] // args.profile_steps
                    avg_bytes, avg_flops = a[2] / a[1], a[3] / a[1]
                    avg_us = fams[dom]["avg_us"]
                    gbs, tfl = avg_bytes / avg_us / 1e3, avg_flops / avg_us / 1e6
                    hbm_bound = (avg_bytes / (HBM_PEAK_GBS * 1e9)) >= (avg_flops / (MFMA_PEAK_TFLOPS * 1e12))
                    tot_us = sum(v["total_us"] for v in fams.values())
                    roofline.update({"kernel": dom, "bound": "hbm" if hbm_bound else "mfma",
                                     "achieved": round(gbs if hbm_bound else tfl, 2), "peak": HBM_PEAK_GBS if hbm_bound else MFMA_PEAK_TFLOPS,
                                     "unit": "GB/s" if hbm_bound else "TFLOP/s",
                                     "frac": round((gbs / HBM_PEAK_GBS) if hbm_bound else (tfl / MFMA_PEAK_TFLOPS), 4),
                                     "avg_launch_us": round(avg_us, 2), "launches_per_step": n_launch,
                                     "alg_bytes_per_launch": int(avg_bytes), "alg_flops_per_launch": int(avg_flops),
                                     "kernel_TFLOPs": round(tfl, 1), "kernel_mfma_frac": round(tfl / MFMA_PEAK_TFLOPS, 4),
                                     "share_of_kernel_time": round(fams[dom]["total_us"] / tot_us, 3),
                                     "timing": "in-step: rocprofv3 --kernel-trace --stats child run of this script, both HIP streams running",
                                     "trace_launches_per_step": round(fams[dom]["calls"] / nst, 1),
                                     "isolated": {k: isolated.get(k) for k in ("kernel", "achieved", "frac", "avg_launch_us", "share_of_kernel_time")},
                                     "in_step_ms_per_step": {k: round(v["total_us"] / nst / 1e3, 3) for k, v in
                                                             sorted(fams.items(), key=lambda kv: -kv[1]["total_us"])[:12]}})
            if not args.no_pmc:
                roofline.update(pmc_for(dom, extra))

    if rank == 0 and roofline is None and work:       # no per-kernel pass (the config 4 child run): the whole-path fraction needs none
        t_window = elapsed / args.steps / B
        roofline = {"kernel": None, "bound": "hbm" if work["bytes"] / (HBM_PEAK_GBS * 1e9) >= work["flop"] / (MFMA_PEAK_TFLOPS * 1e12) else "mfma",
                    "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": None,
                    "frac_whole_path": round(max(work["flop"] / (MFMA_PEAK_TFLOPS * 1e12), work["bytes"] / (HBM_PEAK_GBS * 1e9)) / t_window, 4),
                    "whole_path": {"mfma_frac": round(work["flop"] * wps / world / (MFMA_PEAK_TFLOPS * 1e12), 4),
                                   "hbm_alg_frac_block_granular": round(work["bytes"] * wps / world / (HBM_PEAK_GBS * 1e9), 4),
                                   "definition": "SURVEY.md 8(d): max(F/2.5e15, B/8e12) / t_window; per-kernel fields need --profile-steps > 0"}}
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(max_seconds=args.cpu_seconds, frames=T, frozen=long)
    others = None
    if rank == 0 and world == 1 and args.config == "train" and full and not args.no_other_configs:
        # BASELINE.json configs[3] and configs[4] beside the headline line (child runs of this script, 20 steps / 300 frames)
        torch.cuda.empty_cache()      # (the children are separate processes on the same 288 GB device)
        others = {"long004": other_config("long004", ["--steps", "20", "--warmup", "5", "--cpu-seconds", "8", "--profile-steps", "2", "--trace-only"]),
                  "predict": other_config("predict", ["--steps", "300", "--predict-kernel-trace"])}

    if rank == 0:
        stepk = "torch focal loss / optimizer" if args.torch_step else "mds.train fused focal loss / multi-tensor optimizer"
        what = {"train": f"fwd + focal loss + bwd + grad all-reduce + AdamW ({stepk})",
                "long004": f"frozen 2D encoder fwd (BN in train mode) + tail fwd/bwd + focal loss (alpha 0.4) + grad all-reduce + SGD-Nesterov ({stepk})"}[args.config]
        name = {"train": "sampling_weights_001", "long004": "ball_finetune_long_004"}[args.config]
        out = {"metric": f"frame-windows/sec (fwd+bwd) at {T}x{args.height}x{args.width}, batch {B}", "value": round(wps, 3),
               "unit": "frame-windows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": args.dtype, "data": "synthetic",
               "config": {"workload": f"{name}: {B}x{T}x{args.height}x{args.width} windows/GPU, {what}",
                          "global_batch": B * world, "parallelism": f"dp{world}", "drop_rate": 0.2, "drop_path_rate": 0.2},
               "roofline": roofline, "cpu_baseline": cpu, "kernel_breakdown": breakdown,
               "top_launches": top_launches if args.profile_steps > 0 else None,
               "loss": round(float(loss.detach()), 5), "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 2)}
        if others is not None:
            out["other_configs"] = others
        if par_info is not None:
            out["parallel"] = par_info
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
