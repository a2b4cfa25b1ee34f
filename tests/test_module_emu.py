"""The whole drop-in module (planner + every kernel source) stepped through the host kernel
simulator against the oracle, fp32: logits, loss gradients for every parameter, BN buffers."""
import copy
import os

import numpy as np
import pytest
import torch

from oracle import multidim_stacker_ref as orc
from det_init import fill_deterministic
import mds
from conftest import GOLDEN


def _pair(kw, seed=3, scale=0.05):
    ref = fill_deterministic(orc.MultiDimStacker(**kw), seed, scale=scale)
    prod = mds.MultiDimStacker(**kw)
    prod.load_state_dict(ref.state_dict())
    from hipemu.loader import load_emulator
    prod._lib = load_emulator()
    return ref, prod


def _cmp(name, got, want, rtol, atol_rel):
    got, want = got.detach().float(), want.detach().float()
    err = (got - want).abs().max().item()
    bound = atol_rel * max(want.abs().max().item(), 1e-12) + rtol * want.abs().max().item()
    assert err <= bound, f"{name}: max err {err:.3e} > {bound:.3e} (ref max {want.abs().max().item():.3e})"


def test_state_dict_contract_of_product_module():
    m = mds.MultiDimStacker(**orc.BASIC_CONFIG_KWARGS)
    lines = open(os.path.join(GOLDEN, "state_dict_contract.txt")).read().strip().split("\n")
    assert [f"{k} {tuple(v.shape)}" for k, v in m.state_dict().items()] == lines
    assert sum(p.numel() for p in m.parameters()) == 6_770_547
    m2 = copy.deepcopy(m)                      # src/ema.py:40
    assert m2._cache is not m._cache
    for a in ("num_stacks", "stack_size", "num_3d_features", "num_features", "drop_rate", "conv2d_encoder"):
        assert hasattr(m, a)


def test_product_refuses_cpu_without_hip_library():
    m = mds.MultiDimStacker(**orc.BASIC_CONFIG_KWARGS)
    with pytest.raises(mds.MdsError):
        m(torch.rand(1, 15, 64, 64))


def test_full_model_train_step_fp32_vs_oracle():
    kw = dict(orc.BASIC_CONFIG_KWARGS, drop_rate=0.0, drop_path_rate=0.0)
    ref, prod = _pair(kw)
    ref.train(); prod.train()
    x = torch.rand(1, 15, 48, 40, generator=torch.Generator().manual_seed(1))  # odd sizes down the pyramid
    tgt = torch.tensor([[1.0, 0.0]])
    lr = ref(x)
    orc.sigmoid_focal_loss(lr, tgt, alpha=-1.0, gamma=1.2).backward()
    lp = prod(x)
    orc.sigmoid_focal_loss(lp, tgt, alpha=-1.0, gamma=1.2).backward()
    _cmp("logits", lp, lr, 1e-4, 1e-4)
    rp, pp = dict(ref.named_parameters()), dict(prod.named_parameters())
    worst = []
    # biases that feed a train-mode BatchNorm have an analytically zero gradient (~1e-8 noise):
    # measure every error against max(|ref grad|, 1e-2 * typical grad magnitude)
    floor = 1e-2 * float(np.median([p.grad.abs().max().item() for p in rp.values()]))
    for n in rp:
        g, w = pp[n].grad, rp[n].grad
        assert g is not None, n
        err = (g - w).abs().max().item() / max(w.abs().max().item(), floor)
        worst.append((err, n))
    worst.sort(reverse=True)
    assert worst[0][0] < 2e-3, f"worst relative grad errors: {worst[:8]}"
    for (n, b), (_, b2) in zip(ref.named_buffers(), prod.named_buffers()):
        _cmp("buffer " + n, b2, b, 1e-4, 1e-4)
    # eval mode (running statistics), no grad.  The deterministic fill gives arbitrary running
    # statistics that make the eval network blow up to ~1e5 (ill-conditioned in fp32), so first
    # overwrite them with this batch's statistics (momentum 1) as a trained model would have.
    for bn in ref.modules():
        if isinstance(bn, torch.nn.modules.batchnorm._BatchNorm):
            bn.momentum = 1.0
    with torch.no_grad():
        ref(x)
    prod.load_state_dict(ref.state_dict())
    ref.eval(); prod.eval()
    with torch.no_grad():
        _cmp("eval logits", prod(x), ref(x), 1e-4, 1e-4)


def _step(model, x, tgt):
    model.zero_grad(set_to_none=True)
    logits = model(x)
    orc.sigmoid_focal_loss(logits, tgt, alpha=-1.0, gamma=1.2).backward()
    return logits.detach(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}


@pytest.mark.parametrize("mode", ["0"])
def test_bn_backward_fusion_modes_match_the_default_path(monkeypatch, mode):
    """MDS_FUSE_BN_BWD=0 (every BatchNorm-backward reduce its own launch) must give the gradients of the default path (the sums
    of a block's output BatchNorm taken in the producing data-gradient GEMM's epilogue) - DropPath masks included"""
    kw = dict(orc.BASIC_CONFIG_KWARGS, drop_rate=0.0, drop_path_rate=0.3)
    _, prod = _pair(kw)
    prod.train()
    x = torch.rand(1, 15, 32, 32, generator=torch.Generator().manual_seed(5))
    tgt = torch.tensor([[0.0, 1.0]])
    state = copy.deepcopy(prod.state_dict())

    def run():
        prod.load_state_dict(state)
        prod.clear_plans()
        prod._mask_override = None
        plan = prod._plan(x, "full", 1, 15, 32, 32, True)
        g = torch.Generator().manual_seed(9)
        prod._mask_override = (torch.rand(plan._mask_total, generator=g) < plan.mask_keep).float() / plan.mask_keep
        return _step(prod, x, tgt)

    monkeypatch.delenv("MDS_FUSE_BN_BWD", raising=False)
    l0, g0 = run()
    monkeypatch.setenv("MDS_FUSE_BN_BWD", mode)
    l1, g1 = run()
    prod._mask_override = None
    _cmp("logits", l1, l0, 1e-5, 1e-5)
    floor = 1e-2 * float(np.median([v.abs().max().item() for v in g0.values()]))
    worst = sorted(((g1[n] - g0[n]).abs().max().item() / max(g0[n].abs().max().item(), floor), n) for n in g0)[::-1]
    assert worst[0][0] < 2e-3, f"mode {mode}: worst relative grad differences {worst[:6]}"


def test_torch_compile_wrap_is_one_opaque_call():
    """scripts/ball_action/train.py:83-86 wraps nn_module in torch.compile: the hot path must stay ONE opaque
    call (no tracing into the planner, no recompiles) and give the eager result, forward and backward."""
    import warnings
    import torch._dynamo as dynamo
    kw = dict(orc.BASIC_CONFIG_KWARGS, drop_rate=0.0, drop_path_rate=0.0)
    _, prod = _pair(kw)
    prod.train()
    x = torch.rand(1, 15, 32, 64, generator=torch.Generator().manual_seed(2))
    tgt = torch.tensor([[0.0, 1.0]])
    state = copy.deepcopy(prod.state_dict())
    le, ge = _step(prod, x, tgt)
    prod.load_state_dict(state)
    dynamo.reset()
    cm = torch.compile(prod, backend="inductor")
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        lc, gc = _step(cm, x, tgt)
        prod.load_state_dict(state)
        lc2, gc2 = _step(cm, x, tgt)            # second call: same plan, no recompilation
    msgs = [str(w.message) for w in rec]
    assert not [m for m in msgs if "recompile" in m.lower() or "ctypes" in m.lower()], msgs
    assert torch.equal(lc, le) and torch.equal(lc2, le)
    for n in ge:
        key = n if n in gc else "_orig_mod." + n
        assert torch.equal(gc[key], ge[n]), n
    # the eval/no-grad path and the predictor-style sub-forwards through the compiled wrapper
    cm.eval()
    with torch.no_grad():
        y = cm(x)
        f = cm.forward_2d(x[:, :3])
    assert y.shape == (1, 2) and f.shape[:3] == (1, 1, 192)


def test_eval_mode_backward_uses_running_statistics():
    """forward() in eval() with grad enabled (fine-tuning a frozen-BN model): BatchNorm is an affine map with
    constant running statistics, so dy = gamma*rstd*g (no batch-mean terms) — checked against the oracle."""
    kw = dict(orc.BASIC_CONFIG_KWARGS, drop_rate=0.0, drop_path_rate=0.0)
    ref, prod = _pair(kw)
    x = torch.rand(1, 15, 48, 40, generator=torch.Generator().manual_seed(3))
    tgt = torch.tensor([[1.0, 0.0]])
    for bn in ref.modules():                    # realistic running statistics (see the train-step test)
        if isinstance(bn, torch.nn.modules.batchnorm._BatchNorm):
            bn.momentum = 1.0
    ref.train()
    with torch.no_grad():
        ref(x)
    prod.load_state_dict(ref.state_dict())
    ref.eval(); prod.eval()
    lr, gr = _step(ref, x, tgt)
    lp, gp = _step(prod, x, tgt)
    _cmp("eval logits", lp, lr, 1e-4, 1e-4)
    floor = 1e-2 * float(np.median([g.abs().max().item() for g in gr.values()]))
    worst = sorted(((gp[n] - gr[n]).abs().max().item() / max(gr[n].abs().max().item(), floor), n) for n in gr)[::-1]
    assert worst[0][0] < 2e-3, worst[:6]
    for (n, b), (_, b2) in zip(ref.named_buffers(), prod.named_buffers()):
        assert torch.equal(b.float(), b2.float()), n      # eval: running statistics untouched


def test_second_backward_and_inplace_input_edit_raise():
    kw = dict(orc.BASIC_CONFIG_KWARGS, drop_rate=0.0, drop_path_rate=0.0)
    _, prod = _pair(kw)
    prod.train()
    x = torch.rand(1, 15, 32, 32, generator=torch.Generator().manual_seed(4))
    loss = prod(x).sum()
    loss.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="released"):
        loss.backward()
    x2 = x.clone()
    loss = prod(x2).sum()
    x2.add_(1.0)                                  # the stem weight gradient reads x again in backward
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        loss.backward()


def test_pretrained_true_is_never_a_silent_random_init(monkeypatch):
    """timm.create_model(pretrained=True) fails hard in the reference when the weights are unreachable; so does this
    (ADVICE r2), unless the caller opts in to a random encoder explicitly"""
    monkeypatch.delenv("MDS_ALLOW_RANDOM_INIT", raising=False)
    with pytest.raises(RuntimeError, match="could not be loaded"):
        mds.MultiDimStacker(**dict(orc.BASIC_CONFIG_KWARGS, pretrained=True))
    monkeypatch.setenv("MDS_ALLOW_RANDOM_INIT", "1")
    with pytest.warns(RuntimeWarning, match="RANDOMLY INITIALISED"):
        m = mds.MultiDimStacker(**dict(orc.BASIC_CONFIG_KWARGS, pretrained=True))
    assert m.pretrained_loaded is False


def test_forward_is_a_registered_operator_traceable_with_fullgraph():
    """torch.library.custom_op + register_fake + register_autograd - torch.compile(fullgraph=True) traces through
    forward() (no graph break), eager and compiled results are identical, the EMA-style deep copy is its own operator target,
    frozen parameters get no gradient, and a second backward raises."""
    import torch._dynamo as dynamo
    kw = dict(orc.BASIC_CONFIG_KWARGS, drop_rate=0.0, drop_path_rate=0.0)
    _, prod = _pair(kw)
    prod.train()
    x = torch.rand(1, 15, 32, 64, generator=torch.Generator().manual_seed(21))
    tgt = torch.tensor([[1.0, 0.0]])
    state = copy.deepcopy(prod.state_dict())
    le, ge = _step(prod, x, tgt)
    ema = copy.deepcopy(prod)                      # src/ema.py:40
    assert ema._handle != prod._handle and ema._lib is prod._lib
    prod.load_state_dict(state)
    dynamo.reset()
    cm = torch.compile(prod, fullgraph=True, backend="aot_eager")
    lc, gc = _step(cm, x, tgt)
    assert torch.equal(lc, le)
    for n in ge:
        assert torch.equal(gc[n if n in gc else "_orig_mod." + n], ge[n]), n
    # the graph holds exactly one mds node
    seen = []

    def backend(gm, example_inputs):
        seen.extend(str(n.target) for n in gm.graph.nodes if n.op == "call_function")
        return gm.forward
    dynamo.reset()
    prod.load_state_dict(state)
    torch.compile(prod, fullgraph=True, backend=backend)(x)
    assert sum("mds.forward" in t for t in seen) == 1, seen
    # frozen encoder (src/argus_models.py:104-110): no gradients there, the tail still trains
    for p in prod.conv2d_encoder.parameters():
        p.requires_grad_(False)
    prod.zero_grad(set_to_none=True)
    loss = orc.sigmoid_focal_loss(prod(x), tgt, alpha=-1.0, gamma=1.2)
    loss.backward(retain_graph=True)
    assert all(p.grad is None for p in prod.conv2d_encoder.parameters()) and prod.classifier.weight.grad is not None
    with pytest.raises(RuntimeError, match="released"):
        loss.backward()
    # the copy runs on its own plans (validation uses the EMA copy in eval mode, src/argus_models.py:80-83)
    ema.eval()
    with torch.no_grad():
        assert ema(x).shape == (1, 2)


def test_two_identical_training_calls_in_one_compiled_graph_both_run():
    """a training forward updates the BatchNorm running statistics: two identical calls in one compiled graph (the first one's
    logits unused) must both run - the buffers advance exactly as in eager mode (two momentum updates, num_batches_tracked + 2).
    The operator is tagged nondeterministic_seeded, so graph passes neither merge nor drop it."""
    import torch._dynamo as dynamo
    kw = dict(orc.BASIC_CONFIG_KWARGS, drop_rate=0.0, drop_path_rate=0.0)
    _, prod = _pair(kw)
    prod.train()
    x = torch.rand(1, 15, 32, 64, generator=torch.Generator().manual_seed(5))
    state = copy.deepcopy(prod.state_dict())

    def twice(m, inp):
        m(inp)                    # logits unused
        return m(inp)

    with torch.no_grad():
        ref = twice(prod, x)
    want = copy.deepcopy(prod.state_dict())
    assert int(want["conv2d_encoder.bn1.num_batches_tracked"]) == int(state["conv2d_encoder.bn1.num_batches_tracked"]) + 2
    prod.load_state_dict(state)
    dynamo.reset()
    seen = []

    def backend(gm, example_inputs):
        seen.extend(str(n.target) for n in gm.graph.nodes if n.op == "call_function")
        return gm.forward
    with torch.no_grad():
        got = torch.compile(twice, fullgraph=True, backend="aot_eager")(prod, x)
        dynamo.reset()
        prod.load_state_dict(state)
        got = torch.compile(twice, fullgraph=True, backend=backend)(prod, x)
    assert sum("mds.forward" in t for t in seen) == 2, seen
    assert torch.equal(got, ref)
    have = prod.state_dict()
    for k in want:
        assert torch.equal(have[k], want[k]), k
    # and the writes are visible to version-keyed caches
    v0 = prod.conv2d_encoder.bn1.running_mean._version
    with torch.no_grad():
        prod(x)
    assert prod.conv2d_encoder.bn1.running_mean._version > v0


def test_a_forward_whose_backward_never_runs_releases_its_plan():
    """loss evaluated under grad mode and dropped (or an exception before backward): the plan - a whole activation arena - is
    reusable as soon as the autograd graph dies, not after four later forwards; pickling a module never carries plans"""
    import gc, pickle
    kw = dict(orc.BASIC_CONFIG_KWARGS, drop_rate=0.0, drop_path_rate=0.0)
    _, prod = _pair(kw)
    prod.train()
    x = torch.rand(1, 15, 32, 64, generator=torch.Generator().manual_seed(9))
    out = prod(x)
    assert len(prod._live) == 1
    plan = next(iter(prod._live.values()))
    assert plan.in_flight
    state = prod.__getstate__()
    assert state["_live"] == {} and len(prod._live) == 1
    del out
    gc.collect()
    assert len(prod._live) == 0 and not plan.in_flight
    out = prod(x)                       # the same plan is handed out again
    assert next(iter(prod._live.values())) is plan
    out.sum().backward()
    assert len(prod._live) == 0


def test_plan_cache_is_bounded():
    from mds import module as mod
    kw = dict(orc.BASIC_CONFIG_KWARGS, drop_rate=0.0, drop_path_rate=0.0)
    _, prod = _pair(kw)
    prod.eval()
    with torch.no_grad():
        for w in range(mod.MAX_PLANS + 3):
            prod.forward_head(torch.rand(1, 1280, 1, 1 + w))
    assert prod._cache.count() <= mod.MAX_PLANS
    prod.clear_plans()
    assert prod._cache.count() == 0


def test_pinned_plans_do_not_count_against_the_lru_budget():
    """plans held for a lifetime (a StreamPredictor's) must not make every other shape re-plan on each call"""
    from mds import module as mod
    kw = dict(orc.BASIC_CONFIG_KWARGS, drop_rate=0.0, drop_path_rate=0.0)
    _, prod = _pair(kw)
    prod.eval()
    with torch.no_grad():
        pinned = []
        for w in range(mod.MAX_PLANS + 2):               # more pinned plans than the whole budget
            prod.forward_head(torch.rand(1, 1280, 1, 1 + w))
        for pool in prod._cache.plans.values():
            for p in pool:
                p.in_flight = True
                pinned.append(p)
        x = torch.rand(1, 1280, 2, 3)
        prod.forward_head(x)
        first = [p for pool in prod._cache.plans.values() for p in pool if not p.in_flight]
        prod.forward_head(x)
        again = [p for pool in prod._cache.plans.values() for p in pool if not p.in_flight]
        assert len(first) == 1 and again[0] is first[0], "the idle plan of a repeated shape must be reused, not rebuilt"
        for p in pinned:                                  # released (StreamPredictor.close()): evictable again
            p.in_flight = False
        prod.forward_head(torch.rand(1, 1280, 3, 3))
        assert prod._cache.idle() <= mod.MAX_PLANS


def test_sub_forwards_are_differentiable_and_compose_to_forward():
    """multidim_stacker.py:210-243: forward == forward_head(forward_3d(forward_2d(x))) and every piece is an ordinary
    differentiable method - logits and all parameter gradients of the composition equal those of forward() (and the oracle's)."""
    kw = dict(orc.BASIC_CONFIG_KWARGS, drop_rate=0.0, drop_path_rate=0.0)
    ref, prod = _pair(kw)
    ref.train(); prod.train()
    x = torch.rand(1, 15, 32, 64, generator=torch.Generator().manual_seed(31))
    tgt = torch.tensor([[0.0, 1.0]])
    state = copy.deepcopy(prod.state_dict())
    lf, gf = _step(prod, x, tgt)
    prod.load_state_dict(state)
    prod.zero_grad(set_to_none=True)
    f2 = prod.forward_2d(x)
    assert f2.requires_grad and f2.shape == (1, 5, 192, 1, 2)
    f3 = prod.forward_3d(f2)
    logits = prod.forward_head(f3)
    orc.sigmoid_focal_loss(logits, tgt, alpha=-1.0, gamma=1.2).backward()
    _cmp("composed logits", logits, lf, 1e-6, 1e-6)
    floor = 1e-2 * float(np.median([v.abs().max().item() for v in gf.values()]))
    for n, p in prod.named_parameters():
        assert p.grad is not None, n
        err = (p.grad - gf[n]).abs().max().item() / max(gf[n].abs().max().item(), floor)
        assert err < 2e-3, (n, err)      # (the 3D-tail -> projection seam sums in a different order: fp32 noise on zero-ish BN biases)
    # the oracle composes the same way
    lr, gr = _step(ref, x, tgt)
    _cmp("oracle logits", logits, lr, 1e-3, 1e-4)
    # gradient with respect to the INPUT of forward_3d / forward_head (cached-feature fine-tuning)
    prod.load_state_dict(state)
    feats = f2.detach().clone().requires_grad_(True)
    out = prod.forward_head(prod.forward_3d(feats))
    out.sum().backward()
    fr = f2.detach().clone().requires_grad_(True)
    ref.load_state_dict(state)
    ref.forward_head(ref.forward_3d(fr)).sum().backward()
    _cmp("d/dfeats", feats.grad, fr.grad, 2e-3, 1e-6 + 2e-3 * fr.grad.abs().max().item())
    # under no_grad they are the predictor's inference calls, as before
    with torch.no_grad():
        assert not prod.forward_2d(x[:, :3]).requires_grad
