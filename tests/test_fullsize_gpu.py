"""Full-size checks (BASELINE configs 2 and 4).

Against the oracle, one full window (the oracle's fp32 fwd+bwd of 1 x 15 x 736 x 1280 takes ~3 s on the GPU
box's host, see bench.py cpu_baseline):
  * fp32 HIP vs oracle at 1 x 15 x 736 x 1280: logits, BN buffers and EVERY parameter gradient within 1e-3;
  * config 4 (ball_finetune_long_004): 1 x 33 x 736 x 1280, encoder frozen (fwd only, BN in train mode), tail
    fwd+bwd, fp32, same bar;
  * bf16 HIP vs the fp32 oracle on the same window: logits and the direction/size of the gradient.
And at the full batch of 4 (bf16), through size-independent properties of the HIP path:

  * window-permutation equivariance: BatchNorm statistics are permutation invariant, so permuting the
    windows of the batch permutes the logits and leaves every parameter gradient unchanged;
  * the head is linear in the pooled features: d loss / d classifier.bias == d loss / d logits summed
    over the batch (checked against the autograd of the torch-side loss);
  * a training step with AdamW lowers the loss on the same batch, all gradients finite, running
    statistics updated.
"""
import os

import pytest
import torch

from oracle import multidim_stacker_ref as orc
import mds

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
KW = dict(orc.BASIC_CONFIG_KWARGS, drop_rate=0.0, drop_path_rate=0.0)


def _loss(model, x, tgt):
    with torch.autocast("cuda", dtype=torch.bfloat16):
        logits = model(x)
    return orc.sigmoid_focal_loss(logits.float(), tgt, alpha=-1.0, gamma=1.2), logits


def test_window_permutation_equivariance_full_size():
    torch.manual_seed(0)
    m = mds.MultiDimStacker(**KW).to(DEV).train()
    x = torch.rand(4, 15, 736, 1280, device=DEV, generator=torch.Generator(DEV).manual_seed(5))
    tgt = torch.tensor([[1.0, 0.0], [0.0, 1.0], [1.0, 1.0], [0.0, 0.0]], device=DEV)
    perm = torch.tensor([2, 0, 3, 1], device=DEV)
    state = {k: v.clone() for k, v in m.state_dict().items()}
    m.zero_grad(set_to_none=True)
    l1, y1 = _loss(m, x, tgt); l1.backward()
    g1 = {n: p.grad.clone() for n, p in m.named_parameters()}
    m.load_state_dict(state)                       # same running statistics for the second pass
    m.zero_grad(set_to_none=True)
    l2, y2 = _loss(m, x[perm].contiguous(), tgt[perm]); l2.backward()
    # bf16 storage + atomics in a different order: equal up to accumulation noise
    assert (y2.float() - y1.float()[perm]).abs().max().item() <= 2e-2 * y1.float().abs().max().item() + 1e-3
    assert abs(l1.item() - l2.item()) <= 1e-2 * abs(l1.item()) + 1e-4
    # Gradients: the summation ORDER of the fp32 statistics changes with the permutation, which flips bf16
    # roundings at every layer; cancellation-dominated sums (BatchNorm biases on the residual stream) are
    # then noise (measured: > 100 % on blocks.5.x.bn3.bias, 10 % on the flat vector), exactly as with
    # torch's own bf16 autocast.  Checked: the direction of the whole gradient and the well-conditioned head.
    flat1 = torch.cat([g.flatten() for g in g1.values()])
    flat2 = torch.cat([p.grad.flatten() for _, p in m.named_parameters()])
    cos = torch.nn.functional.cosine_similarity(flat1, flat2, dim=0).item()
    assert cos > 0.98, cos
    gw1, gw2 = g1["classifier.weight"], m.classifier.weight.grad
    assert (gw2 - gw1).abs().max().item() <= 5e-2 * gw1.abs().max().item()


def test_head_bias_gradient_identity_full_size():
    torch.manual_seed(1)
    m = mds.MultiDimStacker(**KW).to(DEV).train()
    x = torch.rand(4, 15, 736, 1280, device=DEV, generator=torch.Generator(DEV).manual_seed(6))
    tgt = torch.tensor([[1.0, 0.0], [0.0, 1.0], [1.0, 1.0], [0.0, 0.0]], device=DEV)
    m.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        logits = m(x)
    logits.retain_grad()
    orc.sigmoid_focal_loss(logits.float(), tgt, alpha=-1.0, gamma=1.2).backward()
    assert torch.allclose(m.classifier.bias.grad, logits.grad.float().sum(0), rtol=1e-4, atol=1e-7)


def test_training_step_lowers_loss_full_size():
    torch.manual_seed(2)
    m = mds.MultiDimStacker(**dict(KW, drop_rate=0.2, drop_path_rate=0.2)).to(DEV).train()
    opt = torch.optim.AdamW(m.parameters(), lr=3e-4)
    x = torch.rand(4, 15, 736, 1280, device=DEV, generator=torch.Generator(DEV).manual_seed(7))
    tgt = torch.tensor([[1.0, 0.0], [0.0, 1.0], [1.0, 0.0], [0.0, 0.0]], device=DEV)
    rm0 = m.conv2d_encoder.bn1.running_mean.clone()
    losses = []
    for _ in range(6):
        opt.zero_grad(set_to_none=True)
        loss, _ = _loss(m, x, tgt)
        loss.backward()
        assert all(torch.isfinite(p.grad).all() for p in m.parameters())
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < losses[0]
    assert not torch.equal(rm0, m.conv2d_encoder.bn1.running_mean)
    assert int(m.conv2d_encoder.bn1.num_batches_tracked) == 6


def test_window_permutation_equivariance_fp32_tight():
    """the same property in fp32 (no autocast) at 4 x 15 x 256 x 320: every gradient must agree tightly,
    which separates implementation errors from the bf16 rounding noise tolerated above"""
    torch.manual_seed(3)
    m = mds.MultiDimStacker(**KW).to(DEV).train()
    x = torch.rand(4, 15, 256, 320, device=DEV, generator=torch.Generator(DEV).manual_seed(8))
    tgt = torch.tensor([[1.0, 0.0], [0.0, 1.0], [1.0, 1.0], [0.0, 0.0]], device=DEV)
    perm = torch.tensor([3, 1, 0, 2], device=DEV)
    state = {k: v.clone() for k, v in m.state_dict().items()}
    m.zero_grad(set_to_none=True)
    y1 = m(x); orc.sigmoid_focal_loss(y1, tgt, alpha=-1.0, gamma=1.2).backward()
    g1 = {n: p.grad.clone() for n, p in m.named_parameters()}
    m.load_state_dict(state)
    m.zero_grad(set_to_none=True)
    y2 = m(x[perm].contiguous()); orc.sigmoid_focal_loss(y2, tgt[perm], alpha=-1.0, gamma=1.2).backward()
    assert torch.allclose(y2, y1[perm], rtol=1e-3, atol=1e-5)
    import numpy as np
    floor = 1e-2 * float(np.median([g.abs().max().item() for g in g1.values()]))
    worst = max((p.grad - g1[n]).abs().max().item() / max(g1[n].abs().max().item(), floor) for n, p in m.named_parameters())
    assert worst < 5e-3, worst


def _full_window_pair(kw, frames, seed):
    from det_init import fill_deterministic
    ref = fill_deterministic(orc.MultiDimStacker(**kw), seed, scale=0.05)
    prod = mds.MultiDimStacker(**kw)
    prod.load_state_dict(ref.state_dict())
    x = torch.rand(1, frames, 736, 1280, generator=torch.Generator().manual_seed(seed + 100))
    return ref, prod.to(DEV), x


def _oracle_step(ref, x, tgt):
    """the oracle in float64: at 294 k - 4.7 M rows per channel the cancellation-dominated sums (BatchNorm biases on
    the residual stream) carry ~1e-3 of fp32 summation noise in ANY fp32 implementation, torch's included — the
    reference value has to be better than the bar it is used for"""
    torch.set_num_threads(min(32, torch.get_num_threads()))
    ref = ref.double()
    ref.zero_grad(set_to_none=True)
    logits = ref(x.double())
    orc.sigmoid_focal_loss(logits, tgt.double(), alpha=-1.0, gamma=1.2).backward()
    return logits.detach().float(), {n: p.grad.detach().float() for n, p in ref.named_parameters() if p.grad is not None}


def _rel(got, want, floor=0.0):
    got, want = got.detach().float().cpu(), want.detach().float().cpu()
    return (got - want).abs().max().item() / max(want.abs().max().item(), floor, 1e-20)


def test_fp32_full_window_vs_oracle_config2():
    """BASELINE configs[1] shape, one window, fp32: HIP path vs the oracle — logits, all 6.77 M gradient values, buffers."""
    import numpy as np
    ref, prod, x = _full_window_pair(KW, 15, 11)
    ref.train(); prod.train()
    tgt = torch.tensor([[1.0, 0.0]])
    lr, gr = _oracle_step(ref, x, tgt)
    prod.zero_grad(set_to_none=True)
    lp = prod(x.to(DEV))
    orc.sigmoid_focal_loss(lp, tgt.to(DEV), alpha=-1.0, gamma=1.2).backward()
    assert _rel(lp, lr) < 1e-3
    gp = {n: p.grad for n, p in prod.named_parameters()}
    floor = 1e-2 * float(np.median([g.abs().max().item() for g in gr.values()]))
    errs = sorted(((_rel(gp[n], gr[n], floor), n) for n in gr), reverse=True)
    # bar 1e-3 (north_star): the cancellation-dominated BatchNorm-bias sums are accumulated in fp64 slots since round 3
    # (with fp32 slot atomics, in a run-dependent order, the worst parameter measured 2e-4 ... 1.05e-3 over runs)
    assert errs[0][0] < 1e-3, errs[:6]
    for (n, b), (_, b2) in zip(ref.named_buffers(), prod.named_buffers()):
        assert _rel(b2, b, 1e-6) < 1e-3, n
    # bf16 kernels on the same window against the fp32 oracle: bf16 rounding through 25 blocks is ~1e-2 on the logits;
    # the gradient must keep its direction and size (a wrong layer in the 6.8 M-vector shows up here, cosine ~0.9 or less)
    prod.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        lb = prod(x.to(DEV))
    orc.sigmoid_focal_loss(lb.float(), tgt.to(DEV), alpha=-1.0, gamma=1.2).backward()
    assert _rel(lb, lr) < 5e-2
    a = torch.cat([p.grad.flatten().cpu() for _, p in prod.named_parameters()])
    b = torch.cat([gr[n].flatten() for n, _ in prod.named_parameters()])
    cos = torch.dot(a, b).item() / (a.norm().item() * b.norm().item())
    # measured 0.989: bf16 storage through 25 blocks; torch's own bf16 autocast sits at the same level (test_module_gpu).
    # A wrong layer is caught by the fp32 comparison above — the bf16 kernels are the same templates.
    assert cos > 0.97 and abs(a.norm().item() / b.norm().item() - 1) < 5e-2, (cos, a.norm().item(), b.norm().item())
    big = sorted(gr, key=lambda n: -gr[n].norm().item())[:10]            # the ten largest gradient tensors, one by one
    gpb = {n: p.grad.float().cpu() for n, p in prod.named_parameters()}
    worst = max(((gpb[n] - gr[n]).norm() / gr[n].norm()).item() for n in big)
    print("bf16, ten largest gradient tensors: worst relative L2 error", worst)
    assert worst < 0.32, worst      # 1.5 x the measured 0.208 (bf16 storage through 25 BatchNorm'd blocks; a mis-scaled layer is > 0.5)


def _host_mem_available_gb():
    with open("/proc/meminfo") as f:
        for line in f:
            if line.startswith("MemAvailable"):
                return int(line.split()[1]) / 2 ** 20
    return 0.0


def test_fp32_and_bf16_batch4_bench_shape_vs_oracle():
    """THE benchmarked configuration itself (BASELINE configs[1]: 4 windows of 15x736x1280, BatchNorm statistics over all 20 stacks):
    fp32 kernels vs the float64 oracle - logits and every gradient value within 1e-3 -, then the bf16 kernels (the bench dtype)
    on the same batch against the same oracle."""
    import numpy as np
    if _host_mem_available_gb() < 400:
        pytest.skip("the float64 oracle at batch 4 needs ~200 GB of host memory")
    from det_init import fill_deterministic
    ref = fill_deterministic(orc.MultiDimStacker(**KW), 21, scale=0.05)
    prod = mds.MultiDimStacker(**KW)
    prod.load_state_dict(ref.state_dict())
    prod = prod.to(DEV)
    ref.train(); prod.train()
    x = torch.rand(4, 15, 736, 1280, generator=torch.Generator().manual_seed(121))
    tgt = torch.tensor([[1.0, 0.0], [0.0, 1.0], [1.0, 1.0], [0.0, 0.0]])
    lr, gr = _oracle_step(ref, x, tgt)
    xd, td = x.to(DEV), tgt.to(DEV)
    prod.zero_grad(set_to_none=True)
    lp = prod(xd)
    orc.sigmoid_focal_loss(lp, td, alpha=-1.0, gamma=1.2).backward()
    assert _rel(lp, lr) < 1e-3
    gp = {n: p.grad for n, p in prod.named_parameters()}
    floor = 1e-2 * float(np.median([g.abs().max().item() for g in gr.values()]))
    errs = sorted(((_rel(gp[n], gr[n], floor), n) for n in gr), reverse=True)
    # the same step by torch itself in fp32 (the oracle's modules, fp32 parameters, eager CPU kernels) against the float64 values:
    # what "fp32" can mean at 20 x 294 400 ... 4.7 M rows per channel
    ref32 = fill_deterministic(orc.MultiDimStacker(**KW), 21, scale=0.05).train()
    ref32.zero_grad(set_to_none=True)
    l32 = ref32(x)
    orc.sigmoid_focal_loss(l32, tgt, alpha=-1.0, gamma=1.2).backward()
    errs32 = sorted(((_rel(p.grad, gr[n], floor), n) for n, p in ref32.named_parameters()), reverse=True)
    print("batch-4 bench shape, worst gradient errors vs float64: HIP fp32", errs[:4], "torch fp32", errs32[:4])
    # bar: 1e-3 (north_star) for EVERY parameter.  Rounds 3-4 needed an allowance for two cancellation-dominated BatchNorm-bias
    # sums (1.2 - 1.5e-3, torch's own fp32 run: 1.8e-3).  Cause, found in round 5: the per-channel means of BatchNorm backward
    # (sum g / M, sum g xhat / M, the batch mean) were fp32 - ONE rounding error shared by all M rows of a channel, so it adds up
    # M times in every sum over dy (the bias gradients upstream).  The fp32 apply pass now subtracts them in fp64
    # (mds_bn_bwd_finalize_args.coef64): measured 3.4e-4 worst.
    assert errs[0][0] < 1e-3, (errs[:6], errs32[:6])
    for (n, b), (_, b2) in zip(ref.named_buffers(), prod.named_buffers()):
        assert _rel(b2, b, 1e-6) < 1e-3, n
    prod.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        lb = prod(xd)
    orc.sigmoid_focal_loss(lb.float(), td, alpha=-1.0, gamma=1.2).backward()
    assert _rel(lb, lr) < 5e-2
    a = torch.cat([p.grad.flatten().cpu() for _, p in prod.named_parameters()])
    b = torch.cat([gr[n].flatten() for n, _ in prod.named_parameters()])
    cos = torch.dot(a, b).item() / (a.norm().item() * b.norm().item())
    print("batch-4 bench shape: fp32 worst gradient error", errs[0], "bf16 logits", _rel(lb, lr), "bf16 gradient cosine", cos)
    assert cos > 0.97 and abs(a.norm().item() / b.norm().item() - 1) < 5e-2, (cos, a.norm().item(), b.norm().item())


def test_fp32_full_window_vs_oracle_config4_frozen_encoder():
    """BASELINE configs[3] (ball_finetune_long_004.py:8,67): num_frames 33, 2D encoder frozen but in train mode."""
    import numpy as np
    kw = dict(KW, num_frames=33)
    ref, prod, x = _full_window_pair(kw, 33, 12)
    for m_ in (ref, prod):
        for p in m_.conv2d_encoder.parameters():
            p.requires_grad_(False)
        m_.train()
    tgt = torch.tensor([[0.0, 1.0]])
    lr, gr = _oracle_step(ref, x, tgt)
    prod.zero_grad(set_to_none=True)
    lp = prod(x.to(DEV))
    orc.sigmoid_focal_loss(lp, tgt.to(DEV), alpha=-1.0, gamma=1.2).backward()
    assert _rel(lp, lr) < 1e-3
    gp = {n: p.grad for n, p in prod.named_parameters() if p.grad is not None}
    assert set(gp) == set(gr) and not any(n.startswith("conv2d_encoder") for n in gp)
    floor = 1e-2 * float(np.median([g.abs().max().item() for g in gr.values()]))
    errs = sorted(((_rel(gp[n], gr[n], floor), n) for n in gr), reverse=True)
    # bar 1e-3 (north_star): the cancellation-dominated BatchNorm-bias sums are accumulated in fp64 slots since round 3
    # (with fp32 slot atomics, in a run-dependent order, the worst parameter measured 2e-4 ... 1.05e-3 over runs)
    assert errs[0][0] < 1e-3, errs[:6]
    for (n, b), (_, b2) in zip(ref.named_buffers(), prod.named_buffers()):
        assert _rel(b2, b, 1e-6) < 1e-3, n          # frozen encoder still updates its running statistics


def test_config4_batch4_properties_with_its_own_recipe():
    """BASELINE configs[3] at its full shape, 4 x 33 x 736 x 1280 bf16, frozen encoder in train mode, the config's own
    recipe (focal alpha 0.4 / gamma 1.2, SGD momentum 0.9 Nesterov at lr 1e-3; ball_finetune_long_004.py:46-55,67):
    window-permutation equivariance of the logits, the bias-gradient identity, no encoder gradients, running statistics of
    the frozen encoder still updated, and the loss goes down over a few fused SGD steps."""
    from mds import train as mtrain
    torch.manual_seed(4)
    kw = dict(orc.BASIC_CONFIG_KWARGS, num_frames=33, drop_rate=0.0, drop_path_rate=0.0)
    m = mds.MultiDimStacker(**kw).to(DEV).train()
    for p in m.conv2d_encoder.parameters():
        p.requires_grad_(False)
    x = torch.rand(4, 33, 736, 1280, device=DEV, generator=torch.Generator(DEV).manual_seed(9))
    tgt = torch.tensor([[1.0, 0.0], [0.0, 1.0], [1.0, 1.0], [0.0, 0.0]], device=DEV)
    loss_fn = mtrain.FocalLoss(alpha=0.4, gamma=1.2)
    state = {k: v.clone() for k, v in m.state_dict().items()}
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y1 = m(x)
    y1.retain_grad()
    l1 = loss_fn(y1, tgt)
    l1.backward()
    assert all(p.grad is None for p in m.conv2d_encoder.parameters())
    assert torch.allclose(m.classifier.bias.grad, y1.grad.float().sum(0), rtol=1e-4, atol=1e-7)
    g1 = torch.cat([p.grad.flatten() for p in m.parameters() if p.grad is not None])
    assert g1.numel() == 1160163 + 2 * (2816 - 1280) and torch.isfinite(g1).all()     # the tail's parameters (classifier 2816 -> 2)
    assert not torch.equal(state["conv2d_encoder.bn1.running_mean"], m.conv2d_encoder.bn1.running_mean)
    perm = torch.tensor([1, 3, 0, 2], device=DEV)
    m.load_state_dict(state)
    m.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y2 = m(x[perm].contiguous())
    loss_fn(y2, tgt[perm]).backward()
    assert (y2.float() - y1.detach().float()[perm]).abs().max().item() <= 2e-2 * y1.detach().float().abs().max().item() + 1e-3
    g2 = torch.cat([p.grad.flatten() for p in m.parameters() if p.grad is not None])
    assert torch.nn.functional.cosine_similarity(g1, g2, dim=0).item() > 0.98
    opt = mtrain.FusedSGD([p for p in m.parameters() if p.requires_grad], lr=1e-3, momentum=0.9, nesterov=True)
    losses = []
    for _ in range(6):
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = loss_fn(m(x), tgt)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < losses[0], losses


def test_bf16_block_outputs_layer_by_layer_vs_fp32_oracle():
    """The benchmarked dtype checked LAYER BY LAYER at the real shape: one 15 x 736 x 1280 window in
    train mode, bf16 kernels, every block output of the plan (27 taps: stage 0 ... stage 5, 2D projection, the four 3D blocks,
    the 3D projection) against the fp32 oracle's output of the same block.  The yardstick is SURVEY 7's criterion applied per
    layer: the relative L2 error of block d must stay within 2x the error torch's own bf16-autocast run of the oracle has at
    that block (bf16 operands through ill-conditioned 3x3 / 1x1 sums cost ~5e-3 per block: measured 6.3e-3 at block 1,
    2.6e-2 at block 9, 0.10 after the 3D projection for the HIP path - torch's own autocast run: 7.2e-3, 2.9e-2, 0.12) - a wrong layer shows up as O(1) from its block on.  The fp32 kernels are checked against the
    same taps at 1e-4, which pins the tap order and layouts themselves."""
    from det_init import fill_deterministic
    ref = fill_deterministic(orc.MultiDimStacker(**KW), 21, scale=0.05).train()
    prod = mds.MultiDimStacker(**KW)
    prod.load_state_dict(ref.state_dict())
    prod = prod.to(DEV).train()
    x = torch.rand(1, 15, 736, 1280, generator=torch.Generator().manual_seed(121))
    torch.set_num_threads(min(32, torch.get_num_threads()))
    sink = []

    def rows2d(t):                       # (N, C, H, W) -> [N*H*W][C]
        return t.detach().float().permute(0, 2, 3, 1).reshape(-1, t.shape[1])

    def rows3d(t):                       # (B, C, T, H, W) -> [B*T*H*W][C]
        return t.detach().float().permute(0, 2, 3, 4, 1).reshape(-1, t.shape[1])
    hooks = []
    for stage in ref.conv2d_encoder.blocks:
        for blk in stage:
            hooks.append(blk.register_forward_hook(lambda m, i, o: sink.append(rows2d(o))))
    hooks.append(ref.conv2d_projection.register_forward_hook(lambda m, i, o: sink.append(rows2d(o))))
    for blk in ref.conv3d_encoder:
        hooks.append(blk.register_forward_hook(lambda m, i, o: sink.append(rows3d(o))))
    hooks.append(ref.conv3d_projection.register_forward_hook(lambda m, i, o: sink.append(rows2d(o))))
    state = {k: v.clone() for k, v in ref.state_dict().items()}
    with torch.no_grad():
        lr = ref(x)
    want = list(sink)
    sink.clear()
    ref.load_state_dict(state)
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
        ref(x)
    torch_bf16 = [((t - w_).norm() / w_.norm()).item() for t, w_ in zip(sink, want)]
    for h in hooks:
        h.remove()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        lp = prod(x.to(DEV))
    plan = next(p for pool in prod._cache.plans.values() for p in pool if p.kind == "full")
    taps = plan.read_taps()
    assert len(taps) == len(want) == 27, (len(taps), len(want))
    report = []
    for (tag, got), w_, et in zip(taps, want, torch_bf16):
        assert got.shape == w_.shape, (tag, got.shape, w_.shape)
        err = ((got.cpu() - w_).norm() / w_.norm()).item()
        report.append((tag, round(err, 4), round(et, 4)))
    for tag, err, et in report:
        assert err < 2 * et + 1e-3 and err < 0.25, (tag, err, et, report)     # (measured: HIP 0.0063 ... 0.103, torch bf16 0.0072 ... 0.121)
    assert _rel(lp, lr) < 5e-2, report
    # the same taps from the fp32 kernels are tight: the structure of the check itself (tap order, layouts) is exact
    with torch.no_grad():
        prod(x.to(DEV))
    plan32 = next(p for pool in prod._cache.plans.values() for p in pool if p.kind == "full" and p.tdt == torch.float32)
    for (tag, got), w_ in zip(plan32.read_taps(), want):
        assert ((got.cpu() - w_).norm() / w_.norm()).item() < 1e-4, tag
    print("per-block relative L2 error (tag, HIP bf16, torch bf16 autocast):", report)


def test_bf16_gradients_tensor_by_tensor_vs_fp32_oracle():
    """The benchmarked dtype's GRADIENTS checked tensor by tensor at the real shape (cosine + "ten largest within
    35 %" would pass a mis-scaled layer).  One 15 x 736 x 1280 window, train mode: the bf16 kernels' gradient of every parameter
    tensor against the fp32 oracle's, with SURVEY 7's yardstick applied per tensor - the relative L2 error must stay within
    BAR x the error torch's own bf16-autocast run of the oracle (same weights, same window, CPU) has on that tensor.  Tensors whose
    gradient is analytically zero (biases in front of a train-mode BatchNorm) or below 1e-4 of the largest tensor norm are
    compared against that floor instead."""
    from det_init import fill_deterministic
    torch.set_num_threads(min(32, torch.get_num_threads()))
    kw = dict(KW, drop_rate=0.0, drop_path_rate=0.0)
    ref = fill_deterministic(orc.MultiDimStacker(**kw), 33, scale=0.05).train()
    prod = mds.MultiDimStacker(**kw)
    prod.load_state_dict(ref.state_dict())
    prod = prod.to(DEV).train()
    x = torch.rand(1, 15, 736, 1280, generator=torch.Generator().manual_seed(133))
    tgt = torch.tensor([[1.0, 0.0]])
    state = {k: v.clone() for k, v in ref.state_dict().items()}

    def grads(autocast):
        ref.load_state_dict(state)
        ref.zero_grad(set_to_none=True)
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
            out = ref(x)
        orc.sigmoid_focal_loss(out.float(), tgt, alpha=-1.0, gamma=1.2).backward()
        return {n: p.grad.detach().float().clone() for n, p in ref.named_parameters()}
    g32, gt = grads(False), grads(True)
    prod.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        lp = prod(x.to(DEV))
    orc.sigmoid_focal_loss(lp.float(), tgt.to(DEV), alpha=-1.0, gamma=1.2).backward()
    gp = {n: p.grad.detach().float().cpu() for n, p in prod.named_parameters()}
    top = max(g.norm().item() for g in g32.values())
    rows = []
    for n, g in g32.items():
        nr = max(g.norm().item(), 1e-4 * top)
        rows.append((n, ((gp[n] - g).norm() / nr).item(), ((gt[n] - g).norm() / nr).item(), g.norm().item() / top))
    worst = sorted(rows, key=lambda r: -(r[1] / (r[2] + 1e-3)))[:8]
    print("bf16 gradients, tensor by tensor (name, HIP error, torch bf16-autocast error, norm / largest norm) - worst ratios:", worst)
    import statistics
    print("median HIP error", statistics.median(r[1] for r in rows), "median torch error", statistics.median(r[2] for r in rows))
    BAR = float(os.environ.get("MDS_TEST_BF16_GRAD_BAR", "2.0"))
    bad = [(n, round(e, 4), round(et, 4)) for n, e, et, _ in rows if e > BAR * et + 2e-2]
    assert not bad, bad[:12]
