"""Ad-hoc GPU check: the bf16 training step at batch sizes 1, 2, 3 (other item / block counts in k_c3.hip than the benchmarked batch 4) with the
row-streaming kernels on (default) and off (MDS_KNOB_C3 = 1): loss and every parameter gradient against each other."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ball-action-spotting_amd")]
import torch
import bench, mds
from mds import cabi, train as mtrain

dev = torch.device("cuda:0")
for B in (1, 2, 3):
    res = {}
    for knob in (0, 1, 2, 3):   # 2 = the default path again (what two runs of the SAME kernels differ by); 3 = the fp32 plan (no autocast): the reference both are judged by
        torch.manual_seed(0)
        model = mds.MultiDimStacker(**dict(bench.CONFIG, drop_rate=0.0, drop_path_rate=0.0)).to(dev).train()
        lib = model._library(next(model.parameters()))
        lib.check(lib.fn["dev_set"](cabi.MDS_KNOB_C3, knob % 2), "dev_set")
        x = torch.rand(B, 15, 736, 1280, device=dev, generator=torch.Generator(dev).manual_seed(1234))
        t = torch.randint(0, 2, (B, 2), device=dev, generator=torch.Generator(dev).manual_seed(4321)).float()
        if knob == 3:
            loss = mtrain.FocalLoss(alpha=-1.0, gamma=1.2)(model(x), t)
        else:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss = mtrain.FocalLoss(alpha=-1.0, gamma=1.2)(model(x), t)
        loss.backward()
        torch.cuda.synchronize()
        res[knob] = (loss.item(), {n: p.grad.float().clone() for n, p in model.named_parameters() if p.grad is not None})
        lib.fn["dev_set"](cabi.MDS_KNOB_C3, 0)
        del model
    (l2, g2) = res[2]
    v0_ = torch.cat([res[0][1][n].flatten() for n in g2]); v2_ = torch.cat([g2[n].flatten() for n in g2])
    print(f"batch {B}: same kernels twice: loss {res[0][0]:.6f} vs {l2:.6f}, whole-gradient relative difference {((v0_ - v2_).norm() / v2_.norm()).item():.3e}")
    gr = res[3][1]
    vr = torch.cat([gr[n].flatten() for n in g2])
    for k_, tag in ((0, "k_c3.hip"), (1, "k_conv.hip only")):
        vk = torch.cat([res[k_][1][n].flatten() for n in g2])
        print(f"batch {B}: bf16 with {tag:16s} against the fp32 plan: loss {res[k_][0]:.6f} vs {res[3][0]:.6f}, whole-gradient relative difference {((vk - vr).norm() / vr.norm()).item():.3e}")
    (l0, g0), (l1, g1) = res[0], res[1]
    v0 = torch.cat([g0[n].flatten() for n in g1]); v1 = torch.cat([g1[n].flatten() for n in g1])
    cos = torch.dot(v0, v1) / (v0.norm() * v1.norm())
    rel = ((v0 - v1).norm() / v1.norm()).item()
    rows = sorted(((((g0[n] - g1[n]).norm() / (g1[n].norm() + 1e-12)).item(), n, g1[n].norm().item(), (g0[n] - g1[n]).norm().item()) for n in g1), reverse=True)[:4]
    print(f"batch {B}: loss {l0:.6f} vs {l1:.6f}; whole gradient: relative difference {rel:.3e}, cosine {cos.item():.6f}; all finite {all(torch.isfinite(v).all().item() for v in g0.values())}")
    for r_, n, nn, dd in rows:
        print(f"     {n:60s} |g| {nn:.3e}  |diff| {dd:.3e}  rel {r_:.2e}")
print("ok")
