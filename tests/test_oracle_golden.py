"""The oracle (oracle/multidim_stacker_ref.py) against golden vectors produced by the
reference's own classes (tests/golden/make_golden.py).  CPU only, fp32."""
import numpy as np
import torch

from oracle import multidim_stacker_ref as orc
from det_init import fill_deterministic, FakeEncoder

TOL = dict(rtol=2e-5, atol=2e-6)


def T(a):
    return torch.from_numpy(np.asarray(a))


def close(a, b, **kw):
    kw = {**TOL, **kw}
    np.testing.assert_allclose(a.detach().numpy() if torch.is_tensor(a) else a, b, **kw)


def test_gem(golden):
    d = golden("gem")
    gem = orc.GeneralizedMeanPooling(3.0)
    x = T(d["x"]).requires_grad_(True)
    y = gem(x)
    (y * T(d["g"])).sum().backward()
    close(y, d["y"]); close(x.grad, d["dx"]); close(gem.p.grad, d["dp"], rtol=1e-4)


def test_se3d(golden):
    d = golden("se3d")
    se = fill_deterministic(orc.SqueezeExcite(16, reduce_ratio=4, act_layer=torch.nn.SiLU), 3)
    x = T(d["x"]).requires_grad_(True)
    y = se(x)
    (y * T(d["g"])).sum().backward()
    close(y, d["y"]); close(x.grad, d["dx"])
    for n, p in se.named_parameters():
        close(p.grad, d["grad." + n], rtol=1e-4, atol=1e-5)


def test_ir3d_train_eval_buffers(golden):
    d = golden("ir3d")
    blk = fill_deterministic(orc.InvertedResidual3d(8, 8, expansion_ratio=3, se_reduce_ratio=4,
                                                    act_layer=torch.nn.SiLU), 6).train()
    x1 = T(d["x1"]).requires_grad_(True)
    y1 = blk(x1)
    (y1 * T(d["g1"])).sum().backward()
    close(y1, d["y1"], rtol=1e-4, atol=1e-5); close(x1.grad, d["dx1"], rtol=1e-4, atol=1e-5)
    for n, p in blk.named_parameters():
        close(p.grad, d["grad1." + n], rtol=2e-4, atol=2e-5)
    for n, b in blk.named_buffers():
        close(b.float(), d["buf1." + n])
    y2 = blk(T(d["x2"]))
    close(y2, d["y2"], rtol=1e-4, atol=1e-5)
    for n, b in blk.named_buffers():
        close(b.float(), d["buf2." + n])
    blk.eval()
    close(blk(T(d["x2"])), d["y_eval"], rtol=1e-4, atol=1e-5)


def _mk(kw_over, seed, scale=0.15):
    kw = dict(orc.BASIC_CONFIG_KWARGS, drop_rate=0.0, drop_path_rate=0.0, **kw_over)
    return fill_deterministic(orc.MultiDimStacker(**kw), seed, scale=scale)


def test_forward_2d_frame_grouping(golden):
    d = golden("fwd2d_grouping")
    orc.ENCODER_REGISTRY["fake_grouping"] = FakeEncoder
    m = _mk(dict(model_name="fake_grouping"), 10).eval()
    with torch.no_grad():
        y = m.forward_2d(T(d["x"]))
    assert y.shape == (2, 5, 192, 2, 3)
    close(y, d["y"], rtol=1e-5, atol=1e-6)


def test_tail_chain(golden):
    d = golden("tail_chain")
    orc.ENCODER_REGISTRY["fake_grouping"] = FakeEncoder
    m = _mk(dict(model_name="fake_grouping"), 10).train()
    feats = T(d["feats"]).requires_grad_(True)
    y3 = m.forward_3d(feats)
    logits = m.forward_head(y3)
    (logits * T(d["g"])).sum().backward()
    close(y3, d["y3"], rtol=1e-4, atol=1e-5)
    close(logits, d["logits"], rtol=1e-4, atol=1e-5)
    close(feats.grad, d["dfeats"], rtol=1e-3, atol=1e-5)
    named = dict(m.named_parameters())
    for k in d:
        if k.startswith("grad."):
            close(named[k[5:]].grad, d[k], rtol=1e-3, atol=1e-5)


def test_full_model_config1(golden):
    """BASELINE.json configs[0]: 15-frame 128x128 stack, batch 1, CPU."""
    d = golden("full_cfg1")
    m = _mk({}, 14, scale=0.02).train()
    logits = m(T(d["x"]))
    loss = orc.sigmoid_focal_loss(logits, T(d["target"]), alpha=-1.0, gamma=1.2)
    loss.backward()
    close(logits, d["logits"], rtol=1e-4, atol=1e-5)
    close(loss, d["loss"], rtol=1e-4)
    named = dict(m.named_parameters())
    for k in d:
        if k.startswith("grad."):
            close(named[k[5:]].grad, d[k], rtol=2e-3, atol=1e-6)
    total = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters())).item()
    np.testing.assert_allclose(total, d["gradnorm_total"], rtol=1e-3)
    close(m.conv2d_encoder.bn1.running_mean, d["buf.conv2d_encoder.bn1.running_mean"])
    m.eval()
    with torch.no_grad():
        close(m(T(d["x"])), d["logits_eval"], rtol=1e-4, atol=1e-5)


def test_structure_param_counts_and_state_dict_contract():
    m = orc.MultiDimStacker(**orc.BASIC_CONFIG_KWARGS)
    enc = sum(p.numel() for p in m.conv2d_encoder.parameters())
    assert enc == 5_610_384                      # timm features-only tf_efficientnetv2_b0
    # + timm's conv_head(192->1280) + bn2 + classifier(1280->1000) == published 7.14 M
    assert enc + 192 * 1280 + 2 * 1280 + 1280 * 1000 + 1000 == 7_139_704
    assert sum(p.numel() for p in m.parameters()) == 6_770_547
    sd = m.state_dict()
    assert len(sd) == 515
    import os
    from conftest import GOLDEN
    lines = open(os.path.join(GOLDEN, "state_dict_contract.txt")).read().strip().split("\n")
    got = [f"{k} {tuple(v.shape)}" for k, v in sd.items()]
    assert got == lines
    assert m.conv2d_encoder.feature_info[4]["num_chs"] == 192


def test_shape_chain_comments():
    """Shape comments of multidim_stacker.py:211-229 at a reduced spatial size."""
    m = orc.MultiDimStacker(**orc.BASIC_CONFIG_KWARGS).eval()
    with torch.no_grad():
        f = m.forward_2d(torch.rand(2, 15, 64, 96))
        assert f.shape == (2, 5, 192, 2, 3)
        y = m.forward_3d(f)
        assert y.shape == (2, 1280, 2, 3)
        assert m.forward_head(y).shape == (2, 2)


def test_encoder_restatement_matches_timm_published_model_card():
    """External pins for the un-vendored timm==0.9.2 encoder: timm's results table lists
    tf_efficientnetv2_b0 with 7.14 M parameters, 0.73 GMACs / 4.77 M activations at 224 px and 0.54 GMACs / 3.51 M activations
    at its 192 px train size.  The features-only restatement + timm's head (conv_head 1x1 192->1280, bn2,
    classifier 1280->1000) must reproduce all five numbers, the stage strides and the feature channel counts."""
    import torch
    from oracle import multidim_stacker_ref as orc
    enc = orc.EfficientNetV2B0Features(in_chans=3).eval()
    params = sum(p.numel() for p in enc.parameters()) + 192 * 1280 + 2 * 1280 + 1280 * 1000 + 1000
    assert params == 7139704

    def profile(size):
        acts, macs, shapes = [0], [0], []

        def hook(m, i, o):
            acts[0] += o.numel()
            if isinstance(m, torch.nn.Conv2d):
                macs[0] += o.numel() * m.in_channels // m.groups * m.kernel_size[0] * m.kernel_size[1]
        hs = [m.register_forward_hook(hook) for m in enc.modules() if isinstance(m, torch.nn.Conv2d)]
        hs += [st.register_forward_hook(lambda m, i, o: shapes.append(tuple(o.shape[1:]))) for st in enc.blocks]
        with torch.no_grad():
            enc(torch.zeros(1, 3, size, size))
        for h in hs:
            h.remove()
        s = size // 32
        return acts[0] + 1280 * s * s + 1000, macs[0] + 192 * 1280 * s * s + 1280 * 1000, shapes
    a224, m224, shapes = profile(224)
    a192, m192, _ = profile(192)
    assert round(a224 / 1e6, 2) == 4.77 and round(a192 / 1e6, 2) == 3.51, (a224, a192)
    # timm's GMACs come from a profiler that also counts normalisation / pooling arithmetic: conv + linear MACs alone are
    # 0.718 G and 0.528 G - within 2 % of the published 0.73 / 0.54
    assert abs(m224 / 1e9 - 0.73) < 0.015 and abs(m192 / 1e9 - 0.54) < 0.015, (m224, m192)
    assert shapes == [(16, 112, 112), (32, 56, 56), (48, 28, 28), (96, 14, 14), (112, 14, 14), (192, 7, 7)]
    assert [f["num_chs"] for f in enc.feature_info] == [16, 32, 48, 112, 192]
