"""SURVEY §8(f) N4 — video frames on the device: the fetcher interface of the reference over an AMD decoder.

``RocDecFrameFetcher`` has the interface and the behaviour of ``src/frame_fetchers/nvdec.py::NvDecFrameFetcher`` /
``src/frame_fetchers/abstract.py::AbstractFrameFetcher``: ``fetch_frame(index=None)`` (sequential or seek + decode, a decode
error is logged and replaced by a zero frame), ``fetch_frames(indexes)`` (one forward sweep from the smallest to the largest
index, intermediate frames decoded and dropped), ``num_frames / width / height / current_index``; frames are (H, W) uint8
luma tensors on ``cuda:<gpu_id>`` - what NVDEC + ``PySurfaceConverter(NV12 -> Y)`` + ``makefromDevicePtrUint8`` deliver.

The decode itself is the platform's video engine: on MI355X that is VCN through **rocDecode**, which is NOT part of this
build's image (no headers, no library, no network) - the fetcher therefore takes a *decoder backend* object and only
implements what is on this side of it: the pitched NV12 surface -> contiguous luma frame step (``mds_frame_luma``, one HBM
streaming copy, straight into a clip tensor or the predictor's frame ring) and the reference's fetch logic.
``open_rocdecode(path, gpu_id)`` builds the backend from rocDecode's Python binding when it is installed and raises with a
clear message otherwise; tests drive the fetcher with a synthetic backend.

Backend protocol (what a rocDecode wrapper has to provide):
    num_frames, width, height : ints
    decode_next() -> Surface            # the next frame in stream order
    seek_and_decode(index) -> Surface   # frame `index`
Surface = (device_pointer: int, pitch_bytes: int, keepalive: object) of an NV12 (or Y-only) surface whose first `height` rows
are the luma plane.
"""
from __future__ import annotations

import ctypes as C
import logging
from pathlib import Path
from typing import Any, Optional

import torch

from . import cabi

logger = logging.getLogger(__name__)


def open_rocdecode(video_path, gpu_id: int):
    """the rocDecode-backed decoder for `video_path`, or a RuntimeError saying what is missing"""
    try:
        import pyRocVideoDecode.decoder as dec        # noqa: F401  (rocDecode's Python binding; absent from the build image)
        import pyRocVideoDecode.demuxer as dmx        # noqa: F401
    except Exception as e:
        raise RuntimeError(
            "mds.frames: rocDecode's Python binding (pyRocVideoDecode) is not installed - video decode on MI355X needs ROCm's "
            f"rocDecode package ({type(e).__name__}: {e}). Pass a decoder backend object to RocDecFrameFetcher instead.") from e
    raise NotImplementedError("mds.frames: the pyRocVideoDecode adapter is not part of this build (the binding was never available to test against)")


class RocDecFrameFetcher:
    """Frames are produced by ONE primitive, `_luma_into(surface, out)`: the decoder's pitched surface is copied (mds_frame_luma)
    straight into the caller's destination - a fresh (H, W) tensor for `fetch_frame`, row k of a preallocated (n, H, W) clip for
    `fetch_frames`, a slot of StreamPredictor's frame ring for `fetch_into`.  There are no per-frame temporaries and no stack /
    concatenate pass: a clip costs n streaming copies into its final place."""

    def __init__(self, video_path: str | Path, gpu_id: int, decoder: Any = None):
        self.video_path = Path(video_path)
        self.gpu_id = gpu_id
        self._dec = decoder if decoder is not None else open_rocdecode(self.video_path, gpu_id)
        self.num_frames = int(self._dec.num_frames)
        self.width = int(self._dec.width)
        self.height = int(self._dec.height)
        self._cursor = -1                 # index of the last frame HANDED OUT (the reference's `_current_index`, nvdec.py:21 / abstract.py:22-24)
        self._lib = None                  # tests inject the kernel simulator here
        self._pending = []                # surfaces whose asynchronous copy may still be running

    @property
    def current_index(self) -> int:
        return self._cursor

    def _device(self):
        return torch.device("cpu") if self._lib is not None else torch.device("cuda", self.gpu_id)

    # ------------------------------------------------------------------ the primitive
    def _luma_into(self, surface: Any, out: torch.Tensor) -> None:
        """luma plane of a pitched NV12 / Y surface -> `out` ((H, W) uint8, contiguous), asynchronously on the current stream"""
        ptr, pitch, keep = surface
        lib = self._lib if self._lib is not None else cabi.load()
        args = cabi.make("mds_frame_luma_args", width=self.width, height=self.height, pitch=int(pitch), count=1, src=int(ptr),
                         surface_stride=0, dst=out)
        stream = torch.cuda.current_stream(out.device).cuda_stream if out.is_cuda else 0
        lib.check(lib.fn["frame_luma"](C.byref(args), stream), "frame_luma")
        # the surface must outlive the asynchronous copy: it is released only once an event recorded behind the copy has fired
        # (a fixed "last 8" window let a 15- / 33-frame clip on a busy stream hand surfaces back to the decoder pool too early)
        ev = None
        if out.is_cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(out.device))
        self._pending.append((keep, out, ev))
        while self._pending and (self._pending[0][2] is None or self._pending[0][2].query()) and len(self._pending) > 1:
            self._pending.pop(0)

    def _produce(self, out: torch.Tensor, at: Optional[int]) -> None:
        """fill `out` with frame `at` (seek + decode) or, for at=None, with the decoder's next frame.  Contract of the reference
        (abstract.py:26-48): the position counter moves only when a frame was decoded, and ANY failure - end of stream, index out
        of range, a corrupt packet, a conversion error - is logged and yields an all-zero frame instead of an exception."""
        try:
            if at is None:
                if self._cursor + 1 >= self.num_frames:
                    raise EOFError("End of frames")
                surface = self._dec.decode_next()
                self._cursor += 1
            else:
                if not 0 <= at < self.num_frames:
                    raise IndexError(f"Frame index {at} out of range")
                surface = self._dec.seek_and_decode(at)
                self._cursor = at
            self._luma_into(surface, out)
        except BaseException as error:
            logger.error(f"Error while fetching frame {at} from '{str(self.video_path)}': {error}."
                         f"Replace by empty frame.")
            out.zero_()

    # ------------------------------------------------------------------ the reference's interface
    def fetch_frame(self, index: Optional[int] = None) -> torch.Tensor:
        out = torch.empty(self.height, self.width, dtype=torch.uint8, device=self._device())
        self._produce(out, index)
        return out

    def fetch_into(self, indexes, out: torch.Tensor) -> torch.Tensor:
        """frames `indexes` (any order, repeats allowed) into out[k] - one forward sweep of the decoder from the smallest to the
        largest index: a seek to the first wanted frame, then sequential decode; frames in between that nobody asked for are
        decoded and dropped (abstract.py:50-68)."""
        assert out.shape == (len(indexes), self.height, self.width) and out.dtype == torch.uint8 and out.is_contiguous()
        rows = {}
        for k, index in enumerate(indexes):
            rows.setdefault(index, []).append(k)
        first, last = min(rows), max(rows)
        for index in range(first, last + 1):
            ks = rows.get(index)
            if ks is None:
                self._dec.decode_next()                      # skipped frame: the decoder moves, nothing is copied
                continue
            self._produce(out[ks[0]], first if index == first else None)
            for k in ks[1:]:
                out[k].copy_(out[ks[0]])
        return out

    def fetch_frames(self, indexes: list[int]) -> torch.Tensor:
        clip = torch.empty(len(indexes), self.height, self.width, dtype=torch.uint8, device=self._device())
        return self.fetch_into(indexes, clip)
