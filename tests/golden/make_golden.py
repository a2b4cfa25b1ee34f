"""Generate tests/golden/*.npz by running THE REFERENCE's own code (build container only).

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference file ``/root/reference/src/models/multidim_stacker.py`` is loaded by path.
Its ``timm`` imports are satisfied by a test-only shim whose content is this repo's own
restatement (oracle/multidim_stacker_ref.py) — timm itself is not installed here.  Classes
``GeneralizedMeanPooling``, ``BatchNormAct3d``, ``SqueezeExcite``, ``InvertedResidual3d`` and
the bodies of ``MultiDimStacker.forward_2d/forward_3d/forward_head/forward`` executed below are
the reference's, line for line.  Only inputs / expected outputs (data) are stored.
"""
import importlib.util
import os
import sys
import types

sys.dont_write_bytecode = True
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from oracle import multidim_stacker_ref as orc  # noqa: E402
from det_init import fill_deterministic, FakeEncoder  # noqa: E402

REF_FILE = "/root/reference/src/models/multidim_stacker.py"


def load_reference():
    orc.ENCODER_REGISTRY["fake_grouping"] = FakeEncoder
    timm = types.ModuleType("timm")
    timm.create_model = orc.create_model
    layers = types.ModuleType("timm.layers")
    layers.DropPath = orc.DropPath
    layers.create_conv2d = orc.create_conv2d
    layers.get_act_layer = orc.get_act_layer
    layers.get_norm_act_layer = orc.get_norm_act_layer
    timm.layers = layers
    sys.modules["timm"] = timm
    sys.modules["timm.layers"] = layers
    spec = importlib.util.spec_from_file_location("ref_multidim_stacker", REF_FILE)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def npy(t):
    return t.detach().cpu().numpy().astype(np.float32)


def gen(seed):
    return torch.Generator().manual_seed(seed)


def main():
    torch.set_num_threads(4)
    ref = load_reference()
    out = {}

    # ---- GeM: fwd, dx, dp (incl. negative inputs -> clamp path) ----
    gem = ref.GeneralizedMeanPooling(3.0)
    x = (torch.randn(2, 6, 5, 7, generator=gen(1)) * 1.5).requires_grad_(True)
    g = torch.randn(2, 6, generator=gen(2))
    y = gem(x)
    (y * g).sum().backward()
    out["gem"] = dict(x=npy(x), g=npy(g), y=npy(y), dx=npy(x.grad), dp=npy(gem.p.grad))

    # ---- 3D SqueezeExcite fwd/bwd ----
    se = fill_deterministic(ref.SqueezeExcite(16, reduce_ratio=4, act_layer=torch.nn.SiLU), 3)
    x = torch.randn(2, 16, 3, 5, 7, generator=gen(4)).requires_grad_(True)
    g = torch.randn(2, 16, 3, 5, 7, generator=gen(5))
    y = se(x)
    (y * g).sum().backward()
    d = dict(x=npy(x), g=npy(g), y=npy(y), dx=npy(x.grad))
    for n, p in se.named_parameters():
        d["grad." + n] = npy(p.grad)
    out["se3d"] = d

    # ---- InvertedResidual3d: two train steps (BN buffers), then eval ----
    blk = fill_deterministic(
        ref.InvertedResidual3d(8, 8, expansion_ratio=3, se_reduce_ratio=4,
                               act_layer=torch.nn.SiLU, drop_path_rate=0.0), 6)
    blk.train()
    x1 = torch.randn(2, 8, 3, 5, 7, generator=gen(7)).requires_grad_(True)
    g1 = torch.randn(2, 8, 3, 5, 7, generator=gen(8))
    y1 = blk(x1)
    (y1 * g1).sum().backward()
    d = dict(x1=npy(x1), g1=npy(g1), y1=npy(y1), dx1=npy(x1.grad))
    for n, p in blk.named_parameters():
        d["grad1." + n] = npy(p.grad)
    for n, b in blk.named_buffers():
        d["buf1." + n] = npy(b.float())
    x2 = torch.randn(2, 8, 3, 5, 7, generator=gen(9))
    y2 = blk(x2)
    d.update(x2=npy(x2), y2=npy(y2))
    for n, b in blk.named_buffers():
        d["buf2." + n] = npy(b.float())
    blk.eval()
    d["y_eval"] = npy(blk(x2))
    out["ir3d"] = d

    # ---- the same reference classes at channel counts the MFMA kernels accept (multiples of 16), so that the
    #      vectors can be fed to the HIP path itself (tests/test_golden_hip.py), not only to the oracle ----
    gem = ref.GeneralizedMeanPooling(3.0)
    x = (torch.randn(3, 16, 5, 7, generator=gen(21)) * 1.5).requires_grad_(True)     # negatives -> clamp path
    g = torch.randn(3, 16, generator=gen(22))
    y = gem(x)
    (y * g).sum().backward()
    out["gem_c16"] = dict(x=npy(x), g=npy(g), y=npy(y), dx=npy(x.grad), dp=npy(gem.p.grad))
    for tag, T_, H_, W_ in (("ir3d_c16_t5", 5, 5, 7), ("ir3d_c16_t3", 3, 6, 5)):
        blk = fill_deterministic(
            ref.InvertedResidual3d(16, 16, expansion_ratio=3, se_reduce_ratio=4,
                                   act_layer=torch.nn.SiLU, drop_path_rate=0.0), 26)
        blk.train()
        x1 = torch.randn(2, 16, T_, H_, W_, generator=gen(27)).requires_grad_(True)
        g1 = torch.randn(2, 16, T_, H_, W_, generator=gen(28))
        y1 = blk(x1)
        (y1 * g1).sum().backward()
        d = dict(x1=npy(x1), g1=npy(g1), y1=npy(y1), dx1=npy(x1.grad))
        for n, p in blk.named_parameters():
            d["grad1." + n] = npy(p.grad)
        for n, b in blk.named_buffers():
            d["buf1." + n] = npy(b.float())
        x2 = torch.randn(2, 16, T_, H_, W_, generator=gen(29))
        y2 = blk(x2)
        d.update(x2=npy(x2), y2=npy(y2))
        for n, b in blk.named_buffers():
            d["buf2." + n] = npy(b.float())
        blk.eval()
        d["y_eval"] = npy(blk(x2))
        out[tag] = d

    # ---- forward_2d grouping with the fake encoder (eval mode) ----
    kw = dict(orc.BASIC_CONFIG_KWARGS, model_name="fake_grouping", drop_rate=0.0, drop_path_rate=0.0)
    m = fill_deterministic(ref.MultiDimStacker(**kw), 10).eval()
    x = torch.rand(2, 15, 64, 96, generator=gen(11))
    with torch.no_grad():
        f = m.forward_2d(x)
    out["fwd2d_grouping"] = dict(x=npy(x), y=npy(f))

    # ---- forward_3d + forward_head chain (train mode, no dropout/droppath) ----
    m.train()
    feats = torch.randn(2, 5, 192, 3, 4, generator=gen(12)).requires_grad_(True)
    g = torch.randn(2, 2, generator=gen(13))
    y3 = m.forward_3d(feats)
    logits = m.forward_head(y3)
    (logits * g).sum().backward()
    out["tail_chain"] = dict(
        feats=npy(feats), g=npy(g), y3=npy(y3), logits=npy(logits), dfeats=npy(feats.grad),
        **{"grad.classifier.weight": npy(m.classifier.weight.grad),
           "grad.global_pool.p": npy(m.global_pool.p.grad),
           "grad.conv3d_projection.0.weight": npy(m.conv3d_projection[0].weight.grad),
           "grad.conv3d_encoder.0.conv_dw.weight": npy(m.conv3d_encoder[0].conv_dw.weight.grad),
           "grad.conv3d_encoder.3.se.conv_reduce.bias": npy(m.conv3d_encoder[3].se.conv_reduce.bias.grad)})

    # ---- whole reference MultiDimStacker at BASELINE config 1 (128x128, batch 1), the encoder
    #      being this repo's restatement (timm absent): pins every line of the reference file ----
    kw = dict(orc.BASIC_CONFIG_KWARGS, drop_rate=0.0, drop_path_rate=0.0)
    m = fill_deterministic(ref.MultiDimStacker(**kw), 14, scale=0.02).train()
    x = torch.rand(1, 15, 128, 128, generator=gen(15))
    tgt = torch.tensor([[1.0, 0.0]])
    logits = m(x)
    loss = orc.sigmoid_focal_loss(logits, tgt, alpha=-1.0, gamma=1.2, reduction="mean")
    loss.backward()
    d = dict(x=npy(x), target=npy(tgt), logits=npy(logits), loss=npy(loss))
    for n in ["classifier.weight", "global_pool.p", "conv2d_encoder.conv_stem.weight",
              "conv2d_encoder.blocks.1.0.conv_exp.weight", "conv2d_encoder.blocks.3.0.conv_dw.weight",
              "conv2d_encoder.blocks.5.7.se.conv_reduce.weight", "conv2d_projection.0.weight",
              "conv3d_encoder.1.bn2.bn3d.weight"]:
        d["grad." + n] = npy(dict(m.named_parameters())[n].grad)
    d["gradnorm_total"] = np.float32(
        torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters())).item())
    d["buf.conv2d_encoder.bn1.running_mean"] = npy(m.conv2d_encoder.bn1.running_mean)
    m.eval()
    with torch.no_grad():
        d["logits_eval"] = npy(m(x))
    out["full_cfg1"] = d

    # state_dict contract (names + shapes) of the reference class built at the basic config
    sd = ref.MultiDimStacker(**orc.BASIC_CONFIG_KWARGS).state_dict()
    with open(os.path.join(HERE, "state_dict_contract.txt"), "w") as f:
        for k, v in sd.items():
            f.write(f"{k} {tuple(v.shape)}\n")

    for name, d in out.items():
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
        print(name, {k: v.shape for k, v in d.items()})


if __name__ == "__main__":
    main()
