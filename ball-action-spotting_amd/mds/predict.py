"""SURVEY §8(f) N1 — sliding-window predictor fast path.

``StreamPredictor`` has the interface and the semantics of the reference's ``MultiDimStackerPredictor.predict``
(``src/predictors.py:50-75``): feed raw frames one at a time with their index; as soon as the 15-frame window
(stride 2) behind a frame is complete it returns ``(sigmoid probabilities averaged over TTA, predict_index)``,
otherwise ``(None, predict_index)``.  What differs is where the work happens:

* frames stay RAW (uint8, e.g. 720x1280) in a device ring; ``PadNormalizeFramesProcessor`` (``src/frames.py:12-66``:
  constant pad to 1280x736, /255) and the kornia ``hflip`` of TTA are folded into the stem kernel's gather
  (``mds_ingest_t``) — no padded fp32 copies, no flipped copy;
* per new frame exactly one stack of 3 frames goes through the 2D encoder (the other four stacks of the window
  were encoded 6, 12, 18, 24 frames earlier); their features stay on the device in the kernels' own channels-last
  layout in a 30-slot store — no NCHW fp32 round trip between ``forward_2d`` and ``forward_3d``;
* eval-mode BatchNorm of all 72 layers is one table-driven launch, and the two launch schedules (2D encoder of one
  stack; 3D tail + head of five) are captured once into hipGraphs: at batch 1-2 the reference's path is bound by
  ~330 kernel launches per frame, a graph replay is one.

The module's weights are read in place (the same ``mds.MultiDimStacker`` instance, e.g. the EMA copy argus loads).
"""
from __future__ import annotations

import torch

from . import cabi


class StackIndexes:
    """src/indexes.py:6-24 (StackIndexesGenerator.make_stack_indexes)"""

    def __init__(self, size: int, step: int):
        self.size, self.step = size, step
        self.behind = (size // 2) * step
        self.ahead = (size - size // 2 - 1) * step

    def make_stack_indexes(self, frame_index: int):
        return list(range(frame_index - self.behind, frame_index + self.ahead + 1, self.step))


class StreamPredictor:
    def __init__(self, nn_module, frame_size=(1280, 736), frame_stack_size: int = 15, frame_stack_step: int = 2,
                 tta: bool = False, use_graphs: bool = True, compute_dtype: str = None):
        self.m = nn_module
        self.m.eval()
        self.compute_dtype = compute_dtype               # None: the module's own rule (fp32 outside autocast, like the reference) | "bf16"
        self.W, self.H = frame_size                      # PadNormalizeFramesProcessor(size=(width, height))
        self.tta = tta
        self.ss = nn_module.stack_size
        self.S = nn_module.num_stacks
        assert frame_stack_size == self.S * self.ss
        self.idx = StackIndexes(frame_stack_size, frame_stack_step)
        self.step = frame_stack_step
        self._predict_offset = self.idx.make_stack_indexes(0)[-1]
        self.span = self.ss * self.step                  # frames between the ends of consecutive stacks of one window
        self.max_chunk = 32
        self.nframes = 2 * self._predict_offset + 1 + self.max_chunk + 3     # raw-frame ring: a window behind every frame of a chunk
        self.nfeat = (self.S - 1) * self.span + self.max_chunk + 8          # feature store: one slot per stack END index (mod)
        self.use_graphs = use_graphs
        self._built = None
        self.encoder_passes = 0          # 2D-encoder passes issued so far (steady state: one per chunk)
        self.reset_buffers()

    def close(self):
        """hand the launch plans back to the module's cache (they stay pinned while the predictor lives)"""
        for c in getattr(self, "plans", {}).values():
            c["g2d"] = c["gtail"] = None
            c["p2d"].in_flight = c["ptail"].in_flight = False
        self.plans = {}
        self.store = None
        self._built = None

    def __del__(self):
        try:
            self.close()
        except Exception:      # interpreter teardown
            pass

    # ------------------------------------------------------------------ state
    def reset_buffers(self):
        self.frame_tag = [None] * self.nframes
        self.feat_tag = [None] * self.nfeat

    def _build(self, frame: torch.Tensor):
        self.close()             # plans of a previous frame shape / device go back to the module's cache (not pinned forever)
        dev = frame.device
        h, w = frame.shape[-2:]
        assert frame.dtype == torch.uint8 and h <= self.H and w <= self.W, "raw uint8 frames no larger than the padded size"
        self.frames = torch.zeros(self.nframes, h, w, dtype=torch.uint8, device=dev)
        self.plans = {}          # chunk size n -> dict(p2d, ptail, graphs, index caches)
        self.store = None
        self._built = (h, w, dev)

    def _chunk(self, n: int):
        """plans / graphs for chunks of n consecutive frames: one 2D-encoder pass over n (x2 with TTA) new stacks,
        one tail pass over n (x2) windows"""
        c = self.plans.get(n)
        if c is not None:
            return c
        h, w, dev = self._built
        m, b = self.m, (2 if self.tta else 1)
        saved = m.compute_dtype
        if self.compute_dtype is not None:
            m.compute_dtype = self.compute_dtype
        try:
            with torch.no_grad():
                probe = self.frames[0]
                p2d = m._plan(probe, "2d", n * b, self.ss, self.H, self.W, False, ingest=(h, w, n))
                ptail = m._plan(probe, "tail", n * b, self.S * self.ss, p2d.h, p2d.w, False)
        finally:
            m.compute_dtype = saved
        p2d.in_flight = ptail.in_flight = True      # owned by this predictor: never handed out to another caller
        f = p2d.h * p2d.w * m.num_3d_features
        if self.store is None:
            self.f, self.tdt = f, p2d.tdt
            self.store = torch.zeros(self.nfeat, b, f, dtype=p2d.tdt, device=dev)      # [slot][orig | flipped][h*w*c]
        c = self.plans[n] = dict(p2d=p2d, ptail=ptail, g2d=None, gtail=None, warm=0, cache={})
        c["2d_w"], c["tail_w"] = p2d.weight_tensors(), ptail.weight_tensors()
        return c

    def _replay(self, c, which):
        """eager for the first calls (kernel attribute opt-ins, allocator warm-up), then one hipGraph replay"""
        plan = c["p2d"] if which == "2d" else c["ptail"]

        # packed weights / eval BatchNorm table: rebuilt only when a parameter or buffer was written since the last pass
        # (tensor version counters - the module's weights are still read in place, a load_state_dict / optimizer step between
        # two frames is picked up by the next one); kept OUT of the replayed graph
        ver = sum(t._version for t in c[which + "_w"])
        if c.get(which + "_ver") != ver:
            plan.refresh_weights()
            c[which + "_ver"] = ver

        def fn():
            plan.begin_forward(None, refresh=False)
            if which == "2d":
                plan.run("f2d")
            else:
                plan.run("f3d"); plan.run("fhead")
        key = "g2d" if which == "2d" else "gtail"
        if not self.use_graphs or plan.device.type != "cuda":
            return fn()
        if c[key] is None:
            if c["warm"] < 4:
                c["warm"] += 1
                return fn()
            g = torch.cuda.CUDAGraph()
            torch.cuda.synchronize(plan.device)
            with torch.cuda.graph(g):
                fn()
            c[key] = g
        c[key].replay()

    def _window_ready(self, index):
        return all(self.frame_tag[i % self.nframes] == i for i in self.idx.make_stack_indexes(index - self._predict_offset))

    def _idx(self, c, key, rows, dev):
        t = c["cache"].get(key)
        if t is None:
            t = c["cache"][key] = torch.tensor(rows, device=dev)
        return t

    # ------------------------------------------------------------------ the reference's API
    @torch.no_grad()
    def predict(self, frame: torch.Tensor, index: int):
        """frame: (h, w) uint8 (any device; moved to the module's device like src/predictors.py:52)"""
        out = self.predict_batch(frame[None], index)
        return out[0]

    @torch.no_grad()
    def predict_batch(self, frames: torch.Tensor, first_index: int):
        """n consecutive frames (n, h, w) uint8 with indexes first_index .. first_index + n - 1 -> the n results
        ``predict`` would return one by one, from ONE pass of the 2D encoder over the n new stacks and ONE pass of the
        tail over the n windows (offline prediction of a whole half has every frame at hand: src/predictors.py is
        latency-bound at batch 1-2)."""
        dev = next(self.m.parameters()).device
        frames = frames.to(device=dev)
        if self._built is None or self._built != (frames.shape[-2], frames.shape[-1], dev):
            self._build(frames[0])
            self.reset_buffers()
        n = frames.shape[0]
        with torch.cuda.device(dev) if dev.type == "cuda" else _Null():
            results, ready = [], []
            for j in range(n):
                index = first_index + j
                slot = index % self.nframes
                self.frame_tag[slot] = index
                results.append((None, index - self._predict_offset))
            # ring update (n <= ring length; the frames of one chunk land in distinct slots)
            assert n <= self.max_chunk, "chunk longer than the frame ring allows"
            if n == 1:
                self.frames[first_index % self.nframes].copy_(frames[0])
            else:
                self.frames[torch.arange(first_index, first_index + n, device=dev) % self.nframes] = frames
            for j in range(n):
                if self._window_ready(first_index + j):
                    ready.append(j)
            if not ready:
                return results
            if len(ready) != n:           # a chunk that straddles the start of the stream: frame by frame for its ready part
                if n == 1:
                    return results
                for j in ready:
                    results[j] = self._run_chunk([first_index + j])[0]
                return results
            for j, r in enumerate(self._run_chunk([first_index + j for j in range(n)])):
                results[j] = r
            return results

    def _run_chunk(self, indexes):
        n = len(indexes)
        c = self._chunk(n)
        p2d, ptail = c["p2d"], c["ptail"]
        dev, b = p2d.device, (2 if self.tta else 1)
        wins = [self.idx.make_stack_indexes(i - self._predict_offset) for i in indexes]
        stacks = [[tuple(w_[s * self.ss:(s + 1) * self.ss]) for s in range(self.S)] for w_ in wins]
        # stacks whose features are not in the store yet, collected ONCE over the whole chunk: windows sit 6 frames apart, so
        # the newest stack of frame j is also the second newest of frame j + 6 of the same chunk - it is encoded once.  In
        # steady state that is exactly the newest stack of every frame: one pass of n stacks.
        uniq = {}
        for sts in stacks:
            for st in sts:
                if self.feat_tag[st[-1] % self.nfeat] != st:
                    uniq.setdefault(st, None)
        pending = list(uniq)
        for r0 in range(0, len(pending), n):           # > 1 pass only right after a (re)start of the stream
            todo = pending[r0:r0 + n]
            todo = todo + [todo[-1]] * (n - len(todo))   # a short last pass repeats its last stack (same features, same slot)
            sel = self._idx(c, ("sel", tuple(st[0] % self.nframes for st in todo)), [i % self.nframes for st in todo for i in st], dev)
            torch.index_select(self.frames, 0, sel, out=p2d.x_u8.tensor.view(n * self.ss, *self.frames.shape[1:]))
            self._replay(c, "2d")
            self.encoder_passes += 1
            fslots = [st[-1] % self.nfeat for st in todo]
            si = self._idx(c, ("fs", tuple(fslots)), fslots, dev)
            if n == 1:
                self.store[fslots[0]].copy_(p2d.feat.tensor.view(b, self.f))
            else:
                self.store[si] = p2d.feat.tensor.view(b, n, self.f).transpose(0, 1)      # images: n originals, then their n mirrored copies
            for st, fs in zip(todo, fslots):
                self.feat_tag[fs] = st
        slots = [[st[-1] % self.nfeat for st in sts] for sts in stacks]
        gi = self._idx(c, ("g", tuple(s_[-1] for s_ in slots)), slots, dev)              # [n][S]
        if b == 1:      # no TTA: [n][S][1][f] is already the tail's [n][1][S][f] - one gather straight into its input
            torch.index_select(self.store.view(self.nfeat, self.f), 0, gi.view(-1), out=ptail.feat.tensor.view(n * self.S, self.f))
        else:
            gathered = self.store[gi]                                                          # [n][S][b][f]
            ptail.feat.tensor.view(n, b, self.S, self.f).copy_(gathered.permute(0, 2, 1, 3))
        self._replay(c, "tail")
        probs = torch.sigmoid(ptail.logits.tensor.view(n, b, -1))                          # nn.Sigmoid, then the TTA mean
        probs = probs[:, 0] if b == 1 else probs.mean(dim=1)
        return [(probs[j], indexes[j] - self._predict_offset) for j in range(n)]


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
