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

* ``predict_stream`` (offline prediction of a whole half - config 5 is throughput-only): the same per-frame (or per-chunk)
  passes, software-pipelined over ``lanes`` internal HIP streams: step j (ring update, encoder pass, tail pass) runs on lane
  j % lanes with that lane's own launch plans and buffers; only the rings are shared, ordered by events.  At batch 1 every
  launch of a pass is a few hundred workgroups at most on 256 CUs and a step is a chain of ~100 dependent launches, so a
  second chain beside the first costs a third of its own time (measured frame by frame, fp32: 1 / 2 / 3 / 4 lanes = 670 /
  1113 / 1426 / 1587 frames/s; a separate stream for the tail passes loses - profiles/LOG.md).  Results reach the caller's
  stream ``lanes`` steps late, behind an event: what the generator yields is safe to use on the current stream.

The module's weights are read in place (the same ``mds.MultiDimStacker`` instance, e.g. the EMA copy argus loads).
"""
from __future__ import annotations

import os

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
    MAX_LANES = 4
    MAX_IN_FLIGHT = 128          # frames of predict_stream steps in flight (lanes x chunk)

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
        self.predict_offset = self._predict_offset      # src/predictors.py:45 (the reference's attribute name)
        self.span = self.ss * self.step                  # frames between the ends of consecutive stacks of one window
        self.max_chunk = 32
        # raw-frame ring: a window behind every frame of a chunk - and, for predict_stream, room for the steps in flight: the
        # ring update of step j must not reach a slot an encoder pass of steps j - lanes + 1 .. j - 1 still reads (those read
        # back to 28 frames behind their first frame): ring > lanes * chunk + 27, lanes * chunk <= MAX_IN_FLIGHT.  Same for the
        # feature store (one slot per stack END index, modulo): a lane's encoder pass runs at most lanes + 1 steps ahead of
        # another lane's tail pass, which reads 24 frames back.
        # The rings start at what predict() / predict_batch need (one chunk in flight on one stream) and GROW on the first
        # predict_stream call that asks for more frames in flight (every predictor paid 128 frames' worth - 0.3 GB).
        self.in_flight = self.max_chunk
        self._size_rings()
        self.use_graphs = use_graphs
        self._built = None
        self.encoder_passes = 0          # 2D-encoder passes issued so far (steady state: one per chunk)
        self.lanes_in_use = 1            # lanes of the last predict_stream (what the device gave: see _lane_streams)
        self._pipe = None                # predict_stream: the streams / lane of the step being issued
        self.reset_buffers()

    def _size_rings(self):
        self.nframes = 2 * self._predict_offset + 1 + self.in_flight + 8
        self.nfeat = (self.S - 1) * self.span + self.in_flight + 2 * self.max_chunk + 8

    def _grow_rings(self, in_flight: int):
        """rings for `in_flight` frames of predict_stream steps in flight: larger rings, every tagged frame / stack moved to its
        slot in them (slot = index modulo the ring length), the cached copy argument blocks dropped (they hold slots and pointers)"""
        in_flight = min(self.MAX_IN_FLIGHT, int(in_flight))
        if in_flight <= self.in_flight:
            return
        old_ft, old_st = self.frame_tag, self.feat_tag
        self.in_flight = in_flight
        self._size_rings()

        def remap(tags, n_new, key):
            best = {}                                   # new slot -> (index, old slot); the newer index wins a collision
            for slot, tag in enumerate(tags):
                if tag is not None:
                    k = key(tag)
                    if k % n_new not in best or best[k % n_new][0] < k:
                        best[k % n_new] = (k, slot)
            new_tags = [None] * n_new
            for ns, (_, os_) in best.items():
                new_tags[ns] = tags[os_]
            return new_tags, [ns for ns in best], [best[ns][1] for ns in best]

        self.frame_tag, fdst, fsrc = remap(old_ft, self.nframes, lambda i: i)
        self.feat_tag, sdst, ssrc = remap(old_st, self.nfeat, lambda st: st[-1])
        if self._built is not None:
            dev = self.frames.device
            new = torch.zeros((self.nframes,) + tuple(self.frames.shape[1:]), dtype=torch.uint8, device=dev)
            if fdst:
                new[torch.tensor(fdst, device=dev)] = self.frames[torch.tensor(fsrc, device=dev)]
            self.frames = new
            if self.store is not None:
                new = torch.zeros((self.nfeat,) + tuple(self.store.shape[1:]), dtype=self.store.dtype, device=dev)
                if sdst:
                    new[torch.tensor(sdst, device=dev)] = self.store[torch.tensor(ssrc, device=dev)]
                self.store = new
            for c in self.plans.values():
                c["cache"] = {}

    def close(self):
        """hand the launch plans back to the module's cache (they stay pinned while the predictor lives)"""
        for c in getattr(self, "plans", {}).values():
            c["graphs"] = {}
            for plan in c["p2d"] + c["ptail"]:
                plan.in_flight = False
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

    def _chunk(self, n: int, lanes: int = 1, tails: int = 1):
        """plans / graphs for chunks of n consecutive frames: one 2D-encoder pass over n (x2 with TTA) new stacks - one plan
        per encoder lane of predict_stream -, one tail pass over n (x2) windows"""
        c = self.plans.get(n)
        if c is not None and len(c["p2d"]) >= lanes and len(c["ptail"]) >= tails:
            return c
        h, w, dev = self._built
        m, b = self.m, (2 if self.tta else 1)
        saved = m.compute_dtype
        if self.compute_dtype is not None:
            m.compute_dtype = self.compute_dtype
        try:
            with torch.no_grad():
                probe = self.frames[0]
                if c is None:
                    c = self.plans[n] = dict(p2d=[], ptail=[], graphs={}, warm={}, ver={}, cache={}, w={})
                while len(c["p2d"]) < lanes:       # a plan that is in flight is never handed out again: every call builds (or finds) another
                    p2d = m._plan(probe, "2d", n * b, self.ss, self.H, self.W, False, ingest=(h, w, n))
                    p2d.in_flight = True           # owned by this predictor
                    c["w"]["2d", len(c["p2d"])] = p2d.weight_tensors()
                    c["p2d"].append(p2d)
                while len(c["ptail"]) < (tails or 1):
                    p2d = c["p2d"][0]
                    pt = m._plan(probe, "tail", n * b, self.S * self.ss, p2d.h, p2d.w, False, ingest=("probs", b))
                    pt.in_flight = True
                    c["w"]["tail", len(c["ptail"])] = pt.weight_tensors()
                    c["ptail"].append(pt)
        finally:
            m.compute_dtype = saved
        p2d = c["p2d"][0]
        f = p2d.h * p2d.w * m.num_3d_features
        if self.store is None:
            self.f, self.tdt = f, p2d.tdt
            self.store = torch.zeros(self.nfeat, b, f, dtype=p2d.tdt, device=dev)      # [slot][orig | flipped][h*w*c]
        self._order_lanes_behind_caller()
        return c

    def _order_lanes_behind_caller(self):
        """Buffers made on the caller's stream (rings, a plan's zero-initialised arenas and tickets) are first used on a lane's
        stream: inside predict_stream the lanes wait for the caller's stream here, so that a zero-fill still queued there cannot
        land on top of a lane's first writes (plans for a new chunk size / a new lane are built while other lanes are running)."""
        if self._pipe is not None:
            ev = torch.cuda.Event()
            ev.record()                  # the current stream here is the caller's (no lane context is open)
            for st in self._streams:
                st.wait_event(ev)

    def _replay(self, c, which, lane=0):
        """eager for the first calls (kernel attribute opt-ins, allocator warm-up), then one hipGraph replay"""
        plan = c["p2d"][lane] if which == "2d" else c["ptail"][lane]
        key = (which, lane)

        # packed weights / eval BatchNorm table: rebuilt only when a parameter or buffer was written since the last pass
        # (tensor version counters - the module's weights are still read in place, a load_state_dict / optimizer step between
        # two frames is picked up by the next one); kept OUT of the replayed graph
        ver = sum(t._version for t in c["w"][key])
        if c["ver"].get(key) != ver:
            plan.refresh_weights()
            c["ver"][key] = ver

        def fn():
            plan.begin_forward(None, refresh=False)
            if which == "2d":
                plan.run("f2d")
            else:
                plan.run("f3d"); plan.run("fhead")
        if not self.use_graphs or plan.device.type != "cuda":
            return fn()
        if c["graphs"].get(key) is None:
            if c["warm"].get(key, 0) < 4:
                c["warm"][key] = c["warm"].get(key, 0) + 1
                return fn()
            g = torch.cuda.CUDAGraph()
            torch.cuda.synchronize(plan.device)
            with torch.cuda.graph(g):
                fn()
            c["graphs"][key] = g
        c["graphs"][key].replay()

    def _window_ready(self, index):
        return all(self.frame_tag[i % self.nframes] == i for i in self.idx.make_stack_indexes(index - self._predict_offset))

    def _rows(self, c, key, dst, dst_pitch, dst_slots, src, src_pitch, src_slots, row_bytes):
        """one mds_copy_rows launch on the current stream: dst row dst_slots[r] <- src row src_slots[r] (pitches in BYTES).  The
        slot numbers are kernel arguments - no index tensor, so nothing is copied from the host; the argument blocks are cached
        per slot pattern (they repeat with the rings' periods)"""
        st = c["cache"].get(key)
        if st is None:
            assert len(dst_slots) == len(src_slots) <= cabi.MDS_COPY_ROWS_MAX
            st = c["cache"][key] = cabi.make("mds_copy_rows_args", dst=dst, src=src, dst_pitch=dst_pitch, src_pitch=src_pitch,
                                              row_bytes=row_bytes, nrows=len(dst_slots), dst_slot=list(dst_slots), src_slot=list(src_slots))
            if len(c["cache"]) > 4096:            # (bounded: a stream restarted at arbitrary indexes makes new patterns)
                c["cache"].pop(next(iter(c["cache"])))
        dev = dst.device
        c["ptail"][0].lib.call("copy_rows", st, torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0)

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
            self._order_lanes_behind_caller()
        n = frames.shape[0]
        with torch.cuda.device(dev) if dev.type == "cuda" else _Null():
            results, ready = [], []
            # the steps still in flight on the OTHER lanes (at most in_flight - n frames of them) read back to 2 * offset frames behind
            # their first frame: a slot written now must not hold a frame that young (the rings are sized for it - _size_rings;
            # this is the check)
            oldest_live = first_index - (max(0, self.in_flight - n) if self._pipe is not None else 0) - 2 * self._predict_offset
            for j in range(n):
                index = first_index + j
                slot = index % self.nframes
                prev = self.frame_tag[slot]
                assert prev is None or prev == index or prev < oldest_live or prev > index, \
                    f"frame ring too short: slot {slot} still holds frame {prev} while frame {index} arrives ({self.nframes} slots)"
                self.frame_tag[slot] = index
                results.append((None, index - self._predict_offset))
            # ring update (n <= ring length; the frames of one chunk land in distinct slots)
            assert n <= self.max_chunk, "chunk longer than the frame ring allows"
            with self._on("enc", frames):
                a = first_index % self.nframes       # consecutive indexes are consecutive ring slots: one copy (two at the wrap)
                if a + n <= self.nframes:
                    self.frames[a:a + n].copy_(frames)
                else:
                    self.frames[a:].copy_(frames[:self.nframes - a])
                    self.frames[:a + n - self.nframes].copy_(frames[self.nframes - a:])
                if self._pipe is not None:       # (the lane waited for the previous step's ring event first: this one covers all earlier frames too)
                    self._pipe["ring"] = torch.cuda.Event()
                    self._pipe["ring"].record()
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

    # ------------------------------------------------------------------ lanes (predict_stream)
    def _on(self, which, *inputs):
        """stream context of a step: the caller's stream, or - inside predict_stream - the stream of the step's lane"""
        if self._pipe is None:
            return _Null()
        st = self._pipe[which]
        for t in inputs:
            if t.is_cuda:
                t.record_stream(st)         # the caller's tensor is read on the internal stream: its memory must outlive that
        return torch.cuda.stream(st)

    def _run_chunk(self, indexes):
        n = len(indexes)
        pipe = self._pipe
        lane = pipe["lane"] if pipe is not None else 0
        c = self._chunk(n, lane + 1, lane + 1)
        p2d, ptail = c["p2d"][lane], c["ptail"][lane]
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
        with self._on("enc"):
            for r0 in range(0, len(pending), n):           # > 1 pass only right after a (re)start of the stream
                todo = pending[r0:r0 + n]
                todo = todo + [todo[-1]] * (n - len(todo))   # a short last pass repeats its last stack (same features, same slot)
                fb = self.frames.shape[1] * self.frames.shape[2]                     # bytes of a raw frame
                src = [i % self.nframes for st in todo for i in st]
                self._rows(c, ("sel", lane, tuple(src)), p2d.x_u8.tensor, fb, range(n * self.ss), self.frames, fb, src, fb)
                self._replay(c, "2d", lane)
                self.encoder_passes += 1
                fslots = [st[-1] % self.nfeat for st in todo]
                eb = self.store.element_size() * self.f                            # bytes of one image's features
                # the pass's images: n originals, then (TTA) their n mirrored copies -> store[slot][orig | flipped]
                self._rows(c, ("fs", lane, b, tuple(fslots)), self.store, eb, [fs * b + bb for bb in range(b) for fs in fslots],
                           p2d.feat.tensor, eb, range(n * b), eb)
                for st, fs in zip(todo, fslots):
                    self.feat_tag[fs] = st
            if pipe is not None:
                enc_done = torch.cuda.Event()
                enc_done.record()
                for st in pending:
                    pipe["stack_ev"][st] = (enc_done, pipe["enc"])
        slots = [[st[-1] % self.nfeat for st in sts] for sts in stacks]
        with self._on("tail"):
            if pipe is not None:       # the features of the window's stacks: encoder passes of this and of earlier steps, on any lane
                me, seen = torch.cuda.current_stream(), set()
                for sts in stacks:
                    for st in sts:
                        ev = pipe["stack_ev"].get(st)
                        if ev is not None and ev[1] != me and id(ev[0]) not in seen:
                            seen.add(id(ev[0]))
                            me.wait_event(ev[0])
            eb = self.store.element_size() * self.f
            # the tail's input [n][b][S][f] <- store[slot of (window j, stack s)][bb]
            self._rows(c, ("g", lane, b, tuple(tuple(s_) for s_ in slots)), ptail.feat.tensor, eb, range(n * b * self.S), self.store, eb,
                       [slots[j][s_] * b + bb for j in range(n) for bb in range(b) for s_ in range(self.S)], eb)
            self._replay(c, "tail", lane)
            probs = ptail.probs.tensor.view(n, -1).clone()          # nn.Sigmoid + the TTA mean came out of the head's launch; the plan's buffer is reused by the next pass
            if pipe is not None:
                pipe["tail_done"] = torch.cuda.Event()
                pipe["tail_done"].record()
                pipe["outs"].append(probs)
        return [(probs[j], indexes[j] - self._predict_offset) for j in range(n)]

    @torch.no_grad()
    def predict_stream(self, frames, first_index: int = 0, chunk: int = 1, lanes: int = None):
        """Generator over an iterable of raw (h, w) uint8 frames with consecutive indexes from ``first_index``: yields, frame by
        frame and in order, exactly what ``predict(frame, index)`` returns - from the same passes (``chunk`` consecutive frames
        per pass, 1 = the reference's pattern: one new stack through the 2D encoder, one window through the tail).  Step j runs
        on lane j % lanes - a HIP stream with its own encoder and tail plans; the rings (raw frames, stack features) are shared:
        a lane's ring update waits for the previous step's, its tail pass for the encoder passes (on other lanes, 6 .. 24 frames
        earlier) that produced the window's older stacks.  The caller's stream is made to wait for step j only after step
        j + lanes has been issued: ``lanes`` steps of look-ahead into ``frames`` (default: 4 lanes for chunks of up to 4 frames,
        3 up to 8, 2 beyond); everything yielded is ordered behind an event on the current stream."""
        dev = next(self.m.parameters()).device
        if not lanes:        # measured (profiles/r05_predict_lanes.txt): bigger passes fill more of the chip themselves
            lanes = 4 if chunk <= 4 else (3 if chunk <= 8 else 2)
        lanes = max(1, min(int(lanes), self.MAX_LANES, self.MAX_IN_FLIGHT // max(1, int(chunk))))
        if dev.type != "cuda":                    # no streams to overlap: plain calls
            idx = first_index
            for fr in _batched(frames, chunk):
                yield from self.predict_batch(torch.stack(list(fr)), idx)
                idx += len(fr)
            return
        with torch.cuda.device(dev):
            self._streams = _lane_streams(dev, self.m._library(next(self.m.parameters())), self.MAX_LANES)
            enc = self._streams
            lanes = min(lanes, len(enc))
            self.lanes_in_use = lanes          # (what bench.py prints beside the rate)
            cur = torch.cuda.current_stream(dev)
            self._grow_rings(lanes * int(chunk))      # (on the caller's stream, before the lanes are ordered behind it)
            for st in self._streams:              # whatever the caller queued so far (weights, earlier predict calls) comes first
                st.wait_stream(cur)
            pending, idx, step, last_ring, stack_ev = [], first_index, 0, None, {}

            def finish(item):
                res, done, outs = item
                if done is not None:
                    cur.wait_event(done)
                    for out in outs:
                        out.record_stream(cur)    # allocated on the tail stream, consumed on the caller's
                return res

            try:
                for fr in _batched(frames, chunk):
                    fr = [f.to(dev, non_blocking=True) for f in fr]
                    batch = fr[0][None] if len(fr) == 1 else torch.stack(fr)
                    ready = torch.cuda.Event()
                    ready.record(cur)             # the frames are complete on the caller's stream here ...
                    lane = enc[step % lanes]
                    lane.wait_event(ready)        # ... and the lane's ring update starts behind that
                    if last_ring is not None:     # ... and behind the ring updates of all earlier steps (made on other lanes)
                        lane.wait_event(last_ring)
                    pipe = self._pipe = dict(enc=lane, tail=lane, lane=step % lanes, ring=None, tail_done=None, outs=[], stack_ev=stack_ev)
                    try:
                        res = self.predict_batch(batch, idx)
                    finally:
                        self._pipe = None
                    last_ring = pipe["ring"]
                    if len(stack_ev) > 64 * chunk + 256:      # events of stacks no window needs any more
                        for st in list(stack_ev)[:len(stack_ev) // 2]:
                            del stack_ev[st]
                    pending.append((res, pipe["tail_done"], pipe["outs"]))
                    idx += len(fr)
                    step += 1
                    while len(pending) > lanes:
                        yield from finish(pending.pop(0))
                while pending:
                    yield from finish(pending.pop(0))
            finally:
                for st in self._streams:          # an abandoned generator leaves nothing running behind the caller's back
                    cur.wait_stream(st)


_LANE_STREAMS = {}       # device -> dict(streams, verified, tries): the lane streams of predict_stream
_LANE_LOG = []           # one line per selection (what bench.py prints; tests read it)


def _own_streams(dev, n):
    """n HIP streams created OUTSIDE torch's stream pool (hipStreamCreateWithFlags, non-blocking), wrapped as ExternalStream: torch
    hands out pool streams round-robin (32 per priority), so a later torch.cuda.Stream() of the caller - or a Plan's side stream -
    could otherwise alias a lane and silently serialise with it."""
    import ctypes
    from .engine import _hip_path
    hip = ctypes.CDLL(_hip_path())
    hip.hipStreamCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
    out = []
    with torch.cuda.device(dev):
        for _ in range(n):
            h = ctypes.c_void_p()
            rc = hip.hipStreamCreateWithFlags(ctypes.byref(h), 1)          # hipStreamNonBlocking
            if rc != 0 or not h.value:
                raise RuntimeError(f"hipStreamCreateWithFlags failed: {rc}")
            out.append(torch.cuda.ExternalStream(h.value, device=dev))
    return out


def _lane_streams(dev, lib, want):
    """The lane streams are chosen per process and device, by measurement.  A HIP stream is bound to one of the runtime's
    hardware queues (GPU_MAX_HW_QUEUES, 4 by default) when it is first used - the least referenced queue at that moment - and
    two lanes on one hardware queue run one after the other (measured at 4 lanes: 1 / 2 / 4 hardware queues -> 695 / 1110 /
    1591 frames/s; 8 queues: 685 - the device serves four at a time).  So: a dozen candidate streams of our own, a copy that
    keeps a quarter of the chip busy for ~0.1 ms issued on two of them at a time - side by side it takes about as long as one,
    on a shared queue twice as long -, and the largest set of candidates that all overlap each other becomes the lanes.
    The choice is then VERIFIED by running all chosen lanes together (they must finish in < 0.88 x the time the same copies take one
    after the other on a single stream; lanes are dropped until they do, with one warning), it is re-measured on the next call (up to three times) when it came out short of
    ``want`` - one noisy measurement on a busy device no longer pins the process to fewer lanes -, ``MDS_PREDICT_LANES=n``
    caps it, and the outcome is on record (``StreamPredictor.lanes_in_use``, ``mds.predict.lane_log()``)."""
    import os
    import time
    import warnings
    cap = int(os.environ.get("MDS_PREDICT_LANES", "0") or 0)
    if cap > 0:
        want = min(want, cap)
    got = _LANE_STREAMS.get(dev)
    if got is not None and (len(got["streams"]) >= want or got["tries"] >= 3):
        return got["streams"][:want]
    tries = (got["tries"] if got else 0) + 1
    cand = got["cand"] if got else _own_streams(dev, 12)
    nb = 48 << 20
    nj = max(want, 2)
    src, dst = torch.empty(nj, nb, dtype=torch.uint8, device=dev), torch.empty(nj, nb, dtype=torch.uint8, device=dev)
    jobs = [cabi.make("mds_copy_rows_args", dst=dst[k], src=src[k], dst_pitch=nb, src_pitch=nb, row_bytes=nb, nrows=1, dst_slot=[0], src_slot=[0])
            for k in range(nj)]

    def timed(streams):
        best = None
        for _ in range(3):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for k, st in enumerate(streams):
                lib.call("copy_rows", jobs[k], st.cuda_stream)
            torch.cuda.synchronize(dev)
            t = time.perf_counter() - t0
            best = t if best is None or t < best else best
        return best
    for st in cand:
        timed([st])                       # first use binds the hardware queue
    n = len(cand)
    best = []
    one = None
    for attempt in range(3):              # (a cold device - clocks still ramping - once gave three lanes where there are four)
        one = min(timed([st]) for st in cand[:4])
        ok = [[False] * n for _ in range(n)]
        for i in range(n):
            for j in range(i + 1, n):
                ok[i][j] = ok[j][i] = timed([cand[i], cand[j]]) < 1.5 * one

        def grow(chosen, start):          # largest set of mutually overlapping candidates (12 candidates: a handful of steps)
            nonlocal best
            if len(chosen) > len(best):
                best = list(chosen)
            if len(best) >= want:
                return
            for k in range(start, n):
                if all(ok[k][c] for c in chosen):
                    grow(chosen + [k], k + 1)
        grow([], 0)
        if len(best) >= want:
            break
    chosen = [cand[k] for k in (best or [0])][:want]
    # all of them together: pairwise overlap does not prove that `want` lanes run side by side
    dropped = 0
    while len(chosen) > 1 and timed(chosen) > 0.88 * timed([chosen[0]] * len(chosen)):      # the same copies one after the other on ONE stream
        # (side by side four of them are bound by memory bandwidth at ~2 x one copy; streams that share a hardware queue take 4 x)
        chosen.pop()
        dropped += 1
    if len(chosen) < want:
        warnings.warn(f"mds.predict: {len(chosen)} of {want} lanes overlap on this device right now "
                      f"({'re-measured on the next stream' if tries < 3 else 'kept'}; MDS_PREDICT_LANES caps the request)", RuntimeWarning)
    _LANE_LOG.append(f"lanes: wanted {want}, chosen {len(chosen)} of {n} candidate streams (pairwise-overlapping set {len(best)}, dropped by the "
                     f"joint run {dropped}); one 48 MB copy {one * 1e6:.0f} us, all lanes together {timed(chosen) * 1e6:.0f} us; attempt {tries}")
    del src, dst, jobs                    # 2 x want x 48 MB of probe buffers go back to the allocator now
    _LANE_STREAMS[dev] = dict(streams=chosen, cand=cand, tries=tries)
    return chosen


def lane_log():
    """what the lane selection measured, one line per (re)selection"""
    return list(_LANE_LOG)


def _batched(iterable, size):
    it = iter(iterable)
    while True:
        out = []
        for _ in range(size):
            try:
                out.append(next(it))
            except StopIteration:
                break
        if not out:
            return
        yield out


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
