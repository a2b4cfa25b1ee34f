import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + "/ball-action-spotting_amd", ROOT + "/tests", ROOT + "/tests/golden"]
import torch
from det_init import fill_deterministic
from oracle import multidim_stacker_ref as orc
import mds
from mds.predict import StreamPredictor
from test_predictor import RefPredictor
kw = dict(orc.BASIC_CONFIG_KWARGS, drop_rate=0.0, drop_path_rate=0.0)
for (size, fshape) in [((96, 64), (58, 90)), ((160, 128), (120, 150)), ((320, 192), (180, 300))]:
    ref = fill_deterministic(orc.MultiDimStacker(**kw), 5, scale=0.05)
    g = torch.Generator().manual_seed(1)
    for bn in ref.modules():
        if isinstance(bn, torch.nn.modules.batchnorm._BatchNorm): bn.momentum = 1.0
    ref.train()
    with torch.no_grad(): ref(torch.rand(1, 15, size[1], size[0], generator=g))
    prod = mds.MultiDimStacker(**kw); prod.load_state_dict(ref.state_dict()); prod = prod.cuda()
    frames = [torch.randint(0, 256, fshape, generator=g, dtype=torch.uint8) for _ in range(33)]
    rp = RefPredictor(ref, size, False)
    refs = [rp.predict(f, i)[0] for i, f in enumerate(frames)]
    for graphs in (False, True):
        sp = StreamPredictor(prod, frame_size=size, tta=False, use_graphs=graphs)
        errs = []
        for i, f in enumerate(frames):
            p, _ = sp.predict(f, i)
            if p is not None: errs.append((p.float().cpu() - refs[i]).abs().max().item())
        print(size, "graphs", graphs, "errs", ["%.1e" % e for e in errs], flush=True)
    # module-level eval forward on the same padded frames for comparison
    x = torch.stack([rp.process(f[None, None])[0, 0] for f in frames[0:29:2]])[None]
    ref.eval(); prod.eval()
    with torch.no_grad():
        e = (torch.sigmoid(prod(x.cuda())).cpu() - torch.sigmoid(ref(x))).abs().max().item()
    print(size, "module eval forward err %.1e" % e, flush=True)
