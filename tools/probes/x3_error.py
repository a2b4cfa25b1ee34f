"""Developer probe (GPU box): error of an fp32 INFERENCE forward against the float64 oracle - run once per library build
(split-bf16 products, MDS_EVAL_X3=1, against the exact fp32 MFMA build) and compare the printed numbers.
   python tools/probes/x3_error.py [H W]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ball-action-spotting_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import torch
from oracle import multidim_stacker_ref as orc
from det_init import fill_deterministic
import mds

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (368, 640)
torch.set_num_threads(32)
kw = dict(orc.BASIC_CONFIG_KWARGS, drop_rate=0.0, drop_path_rate=0.0)
ref = fill_deterministic(orc.MultiDimStacker(**kw), 5, scale=0.02)
g = torch.Generator().manual_seed(0)
ref.train()
with torch.no_grad():
    for _ in range(4):          # running statistics from the frame distribution (a random eval network is ill-conditioned)
        ref(torch.rand(1, 15, H // 2, W // 2, generator=g))
ref.eval()
prod = mds.MultiDimStacker(**kw)
prod.load_state_dict(ref.state_dict())
prod = prod.to("cuda:0").eval()
x = torch.rand(2, 15, H, W, generator=g)
with torch.no_grad():
    r64 = ref.double()
    l64 = r64(x.double())
    l32 = fill_deterministic(orc.MultiDimStacker(**kw), 5, scale=0.02)
    l32.load_state_dict({k: v.float() for k, v in r64.state_dict().items()})
    l32 = l32.eval()(x)
    lp = prod(x.to("cuda:0")).cpu()
with torch.no_grad():        # the 2D encoder alone (23x40x192 features of one stack of three frames): far less contractive than the logits
    fr = x[:, :3].contiguous()
    f64 = r64.forward_2d(fr.double())
    fp = prod.forward_2d(fr.to("cuda:0")).cpu()
rel = lambda a, b: ((a.double() - b).abs().max() / b.abs().max()).item()
print(f"{H}x{W}: forward_2d features max|ref| {f64.abs().max().item():.4g}, HIP vs float64 {rel(fp, f64):.3e} (rms {((fp.double() - f64).pow(2).mean().sqrt() / f64.pow(2).mean().sqrt()).item():.3e})")
print(f"{H}x{W}: logits max|ref| {l64.abs().max().item():.4g}; HIP fp32 inference plan vs float64 {rel(lp, l64):.3e}; torch fp32 vs float64 {rel(l32, l64):.3e}")
