"""Data-parallel path on CPU: gloo, world_size 2 (the GPU path is the same code over RCCL)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mds import parallel


def test_shard_windows_partitions_everything():
    for n, w in [(6000, 8), (7, 3), (5, 8), (16, 2)]:
        parts = [list(parallel.shard_windows(n, r, w)) for r in range(w)]
        assert sorted(sum(parts, [])) == list(range(n))
        assert max(map(len, parts)) - min(map(len, parts)) <= 1


def test_scaling_model_of_the_bench_line():
    """bench.py --gpus N prints parallel.predicted_ms from this model next to the measurement"""
    slices = [6.0e6, 6.2e6, 6.1e6, 6.4e6, 2.4e6]                # the five slices of the 27.1 MB arena, backward order
    assert parallel.ring_allreduce_ms(27.1e6, 1) == 0.0
    t8 = parallel.ring_allreduce_ms(27.1e6, 8)                  # 14 steps of 3.39 MB over one 153 GB/s link + 14 hops
    assert abs(t8 - (14 * (27.1e6 / 8 / 153e9 * 1e3 + 0.006))) < 1e-9 and 0.3 < t8 < 0.45
    for n, lo in ((2, 0.985), (4, 0.975), (8, 0.97)):
        m = parallel.predict_step_ms(12.6, n, slices)
        assert m["predicted_ms"] > 12.6 and m["predicted_efficiency"] >= lo, (n, m)
        assert m["predicted_exposed_ms"] == round(parallel.ring_allreduce_ms(slices[-1], n), 4)
    assert parallel.predict_step_ms(12.6, 1, slices)["predicted_ms"] == 12.6
    # a paused sync (bench: the N = 1 time measured on the same device) exchanges nothing
    s = parallel.BucketedSync(force=True)
    s.paused = True
    assert not s.active()


def _worker(rank, world, port, tmp, frozen=False, gpu=False):
    for p in sys.path_extra:
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    import mds
    from mds import parallel as par
    from oracle import multidim_stacker_ref as orc
    from det_init import fill_deterministic
    dev = torch.device("cuda:0" if gpu else "cpu")       # gpu: both ranks share the box's one MI355X (gloo moves the slices)

    # 1. flat-buffer mean all-reduce
    flat = torch.full((1000,), float(rank + 1), device=dev)
    par.allreduce_mean_(flat)
    assert torch.allclose(flat.cpu(), torch.full((1000,), (1 + world) / 2 * 1.0))

    # 2. whole module: rank-local windows + local BN, averaged gradients == mean of the oracle's
    kw = dict(orc.BASIC_CONFIG_KWARGS, drop_rate=0.0, drop_path_rate=0.0)
    prod = mds.MultiDimStacker(**kw)
    fill_deterministic(prod, 3 + rank, scale=0.05)          # ranks start different on purpose
    if gpu:
        prod = prod.to(dev)                                  # the product library (libmds_hip.so), both plan streams + the comm stream
    else:
        from hipemu.loader import load_emulator
        prod._lib = load_emulator()
    par.data_parallel(prod)                                  # broadcasts rank 0's state, installs the sync
    ref = orc.MultiDimStacker(**kw)
    ref.load_state_dict({k: v.cpu() for k, v in prod.state_dict().items()})
    ref0 = fill_deterministic(orc.MultiDimStacker(**kw), 3, scale=0.05)
    for a, b in zip(ref.state_dict().values(), ref0.state_dict().values()):
        assert torch.equal(a, b)                             # the packed broadcast made every rank equal to rank 0
    if frozen:                                               # config 4: only the temporal tail's 1.16 M parameters are exchanged
        for m_ in (prod, ref0):
            for p_ in m_.conv2d_encoder.parameters():
                p_.requires_grad_(False)
    # (64x32: at 32x32 the last stages see 1x1 maps, BN over 5 samples is ill-conditioned in fp32)
    xs = [torch.rand(1, 15, 64, 32, generator=torch.Generator().manual_seed(10 + r)) for r in range(world)]
    tgt = torch.tensor([[1.0, 0.0]])
    prod.train()
    orc.sigmoid_focal_loss(prod(xs[rank].to(dev)), tgt.to(dev), alpha=-1.0, gamma=1.2).backward()
    want = None
    for r in range(world):                                   # oracle on every shard, fresh copy each
        m = orc.MultiDimStacker(**kw); m.load_state_dict(ref0.state_dict()); m.train()
        if frozen:
            for p_ in m.conv2d_encoder.parameters():
                p_.requires_grad_(False)
        orc.sigmoid_focal_loss(m(xs[r]), tgt, alpha=-1.0, gamma=1.2).backward()
        g = torch.cat([p.grad.flatten() for p in m.parameters() if p.grad is not None])
        want = g if want is None else want + g
    want /= world
    got = torch.cat([p.grad.flatten() for p in prod.parameters() if p.grad is not None]).cpu()
    assert got.numel() == want.numel() and (not frozen or all(p.grad is None for p in prod.conv2d_encoder.parameters()))
    err = (got - want).abs().max().item() / want.abs().max().item()
    assert err < 1e-3, err
    # the exchange ran as several slices of the flat arena, in backward order (head first, stem last), covering it once
    buckets = prod._grad_sync.buckets
    total = sum(p.numel() for p in prod.parameters())
    if frozen:       # the frozen encoder's part of the arena [0, 5 610 384) is never exchanged: one slice, the tail's 1.16 M elements
        lo_tail = sum(p.numel() for p in prod.conv2d_encoder.parameters())
        assert buckets[0][1] == total and buckets[-1][0] == lo_tail and sum(hi - lo for lo, hi in buckets) == total - lo_tail == got.numel()
    else:
        assert len(buckets) >= 3 and buckets[0][1] == total and buckets[-1][0] == 0
        assert all(hi - lo >= 1_000_000 for lo, hi in buckets[:-1])
    assert all(a[0] == b[1] for a, b in zip(buckets, buckets[1:]))
    torch.save(got, os.path.join(tmp, f"g{rank}.pt"))
    dist.barrier()
    if rank == 0:                                            # every rank holds the same averaged gradient
        g1 = torch.load(os.path.join(tmp, "g1.pt"))
        assert torch.equal(got, g1)
    dist.destroy_process_group()


@pytest.mark.parametrize("world,frozen", [(2, False), (4, False), (2, True)])
def test_gloo_gradients_match_mean_of_oracle(tmp_path, world, frozen):
    sys.path_extra = [p for p in sys.path if "repo" in p]
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    procs = []
    for r in range(world):
        p = ctx.Process(target=_spawn_entry, args=(r, world, port, str(tmp_path), sys.path_extra, frozen))
        p.start(); procs.append(p)
    for p in procs:
        p.join(900)
        assert p.exitcode == 0


def _spawn_entry(rank, world, port, tmp, paths, frozen=False, gpu=False):
    sys.path_extra = paths
    _worker(rank, world, port, tmp, frozen, gpu)


@pytest.mark.gpu
@pytest.mark.parametrize("frozen", [False, True])
def test_world2_sharing_one_gpu_gradients_match_mean_of_oracle(tmp_path, frozen):
    """world size 2 on the HIP path itself: two processes, each with its own window shard, both on the box's single MI355X
    (RCCL refuses two ranks on one device, so the slices travel over gloo): the plan's two streams, the communication stream's
    waits and the in-place sliced all-reduce run for real; averaged gradients == mean of the oracle's per-shard gradients"""
    if not torch.cuda.is_available():
        pytest.skip("needs the GPU")
    sys.path_extra = [p for p in sys.path if "repo" in p]
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    procs = []
    for r in range(2):
        p = ctx.Process(target=_spawn_entry, args=(r, 2, port, str(tmp_path), sys.path_extra, frozen, True))
        p.start(); procs.append(p)
    for p in procs:
        p.join(600)
        assert p.exitcode == 0


@pytest.mark.gpu
def test_bench_line_of_two_ranks_runs_end_to_end(tmp_path):
    """`python bench.py --gpus 2` end to end on the one-GPU box (MDS_BENCH_SHARED_GPU=1: both ranks on device 0, slices over gloo -
    a FUNCTIONAL run of the N > 1 code path, not a measurement): the self-spawned torchrun, the timed steps with the bucketed
    exchange, the steps without it, the rank-0-only per-kernel pass (which must not enter a collective) and the `parallel` block
    with the scaling model's prediction."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MDS_BENCH_SHARED_GPU="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--batch", "1",
                        "--no-cpu-baseline", "--no-pmc", "--no-other-configs"], env=env, capture_output=True, text=True, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
    d = json.loads(lines[0])
    par = d["parallel"]
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and par["rccl_ranks"] == 2 and par["devices_in_use"] == 1
    assert par["predicted_ms"] > par["local_step_ms_no_exchange"] > 0 and 0 < par["efficiency_vs_n1"] <= 1.05
    assert len(par["allreduce_slices_in_backward_order"]) >= 3 and d["roofline"]["kernel"]
