"""SURVEY 8(f) N1 — mds.predict.StreamPredictor against the reference's predictor LOGIC (src/predictors.py:50-75,
src/frames.py:12-66, src/indexes.py:6-24) restated around the oracle model: same (prediction, predict_index) stream
for raw uint8 frames, with and without horizontal-flip TTA."""
import pytest
import torch
import torch.nn.functional as F

from backends import be  # noqa: F401
from det_init import fill_deterministic
from oracle import multidim_stacker_ref as orc
import mds
from mds.predict import StreamPredictor, StackIndexes


class RefPredictor:
    """the reference's MultiDimStackerPredictor.predict, line by line, on an oracle nn_module (no argus)"""

    def __init__(self, nn_module, size, tta):
        self.m, self.size, self.tta = nn_module.eval(), size, tta
        self.gen = StackIndexes(15, 2)
        self.offset = self.gen.make_stack_indexes(0)[-1]
        self.frames, self.feats = {}, {}

    def process(self, frames):      # PadNormalizeFramesProcessor(size=(W, H))
        h, w = frames.shape[-2:]
        hp, wp = self.size[1] - h, self.size[0] - w
        frames = F.pad(frames, [wp // 2, wp - wp // 2, hp // 2, hp - hp // 2], mode="constant", value=0)
        return frames.to(torch.float32) / 255.0

    @torch.no_grad()
    def predict(self, frame, index):
        self.frames[index] = self.process(frame[None, None, ...])[0, 0]
        pi = index - self.offset
        idxs = self.gen.make_stack_indexes(pi)
        for k in [k for k in self.frames if k < idxs[0]]:
            del self.frames[k]
        if not set(idxs) <= set(self.frames):
            return None, pi
        stacks = [tuple(idxs[i:i + 3]) for i in range(0, 15, 3)]
        for st in stacks:
            if st not in self.feats:
                fr = torch.stack([self.frames[i] for i in st], dim=0)
                fr = torch.stack([fr, torch.flip(fr, dims=[-1])], dim=0) if self.tta else fr.unsqueeze(0)
                self.feats[st] = self.m.forward_2d(fr)
        feats = torch.cat([self.feats[s] for s in stacks], dim=1)
        pred = torch.sigmoid(self.m.forward_head(self.m.forward_3d(feats)))
        return pred.mean(dim=0), pi


@pytest.mark.parametrize("tta", [False, True])
def test_stream_predictor_matches_reference_logic(be, tta):
    kw = dict(orc.BASIC_CONFIG_KWARGS, drop_rate=0.0, drop_path_rate=0.0)
    ref = fill_deterministic(orc.MultiDimStacker(**kw), 5, scale=0.05)
    g = torch.Generator().manual_seed(1)
    size = (96, 64)                                            # (width, height) like configs' frames_processor
    # Every frame = one base picture + small noise, and the running statistics come from a window of such frames: eval
    # mode is then as well conditioned as on real footage.  (With arbitrary running statistics the eval network amplifies
    # 1-ulp differences of exp/rcp between CPU and GPU to 1e-2 on the probabilities — measured.)
    base = torch.randint(20, 236, (58, 90), generator=g)

    def new_frame():
        return (base + torch.randint(-12, 13, (58, 90), generator=g)).clamp(0, 255).to(torch.uint8)
    for bn in ref.modules():
        if isinstance(bn, torch.nn.modules.batchnorm._BatchNorm):
            bn.momentum = 1.0
    ref.train()
    rp0 = RefPredictor(ref, size, tta)
    with torch.no_grad():
        ref(torch.stack([rp0.process(new_frame()[None, None])[0, 0] for _ in range(15)])[None])
    prod = mds.MultiDimStacker(**kw)
    prod.load_state_dict(ref.state_dict())
    prod = prod.to(be.device)
    if be.name == "emu":
        prod._lib = be.lib
    rp = RefPredictor(ref, size, tta)
    sp = StreamPredictor(prod, frame_size=size, tta=tta)
    n = 31 if be.name == "emu" else 48
    got, all_frames, refs = 0, [], []
    for index in range(n):
        frame = new_frame()                                   # smaller than the padded size: real padding
        all_frames.append(frame)
        pr, ir = rp.predict(frame, index)
        refs.append(pr)
        pp, ip = sp.predict(frame, index)
        assert ir == ip
        assert (pr is None) == (pp is None), index
        if pr is not None:
            got += 1
            assert pp.shape == pr.shape
            err = (pp.float().cpu() - pr).abs().max().item()
            assert err < 2e-3, (index, err, pr, pp)
    assert got == n - 28
    live = torch.stack([r for r in refs if r is not None])
    assert ((live > 0.01) & (live < 0.99)).any(), "saturated probabilities would make this comparison vacuous"
    # chunked offline prediction: the same stream three frames at a time (one 2D pass over 3 new stacks, one tail pass
    # over 3 windows), including the chunk that straddles the first complete window
    sb = StreamPredictor(prod, frame_size=size, tta=tta)
    for first in range(0, n - n % 3, 3):
        outs = sb.predict_batch(torch.stack(all_frames[first:first + 3]), first)
        for j, (pp, ip) in enumerate(outs):
            assert ip == first + j - 14 and (pp is None) == (refs[first + j] is None)
            if pp is not None:
                assert (pp.float().cpu() - refs[first + j]).abs().max().item() < 2e-3
    # a gap in the stream: the window is incomplete again until 15 fresh frames (stride 2) are there
    sp2 = StreamPredictor(prod, frame_size=size, tta=tta)
    for index in list(range(0, 30)) + [40]:
        out, _ = sp2.predict(torch.zeros(58, 90, dtype=torch.uint8), index)
    assert out is None
