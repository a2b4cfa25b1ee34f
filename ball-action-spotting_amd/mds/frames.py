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
    def __init__(self, video_path: str | Path, gpu_id: int, decoder: Any = None):
        self.video_path = Path(video_path)
        self.gpu_id = gpu_id
        self._dec = decoder if decoder is not None else open_rocdecode(self.video_path, gpu_id)
        self.num_frames = int(self._dec.num_frames)
        self.width = int(self._dec.width)
        self.height = int(self._dec.height)
        self._current_index = -1          # (VPF "skips the first frame at start": nvdec.py:21 starts at 0; a backend states its own origin)
        self._lib = None                  # tests inject the kernel simulator here

    # ------------------------------------------------------------------ src/frame_fetchers/abstract.py, restated
    @property
    def current_index(self) -> int:
        return self._current_index

    def _device(self):
        return torch.device("cpu") if self._lib is not None else torch.device("cuda", self.gpu_id)

    def fetch_frame(self, index: Optional[int] = None) -> torch.Tensor:
        try:
            if index is None:
                if self._current_index < self.num_frames - 1:
                    frame = self._next_decode()
                    self._current_index += 1
                else:
                    raise RuntimeError("End of frames")
            else:
                if index < 0 or index >= self.num_frames:
                    raise RuntimeError(f"Frame index {index} out of range")
                frame = self._seek_and_decode(index)
                self._current_index = index
            frame = self._convert(frame)
        except BaseException as error:      # abstract.py:40-48: a broken frame becomes an empty one, the stream goes on
            logger.error(f"Error while fetching frame {index} from '{str(self.video_path)}': {error}."
                         f"Replace by empty frame.")
            frame = torch.zeros(self.height, self.width, dtype=torch.uint8, device=self._device())
        return frame

    def fetch_frames(self, indexes: list[int]) -> torch.Tensor:
        min_frame_index, max_frame_index = min(indexes), max(indexes)
        index2frame = dict()
        frame_indexes_set = set(indexes)
        for index in range(min_frame_index, max_frame_index + 1):
            if index not in frame_indexes_set:
                self._next_decode()
                continue
            index2frame[index] = self.fetch_frame(index) if index == min_frame_index else self.fetch_frame()
        return torch.stack([index2frame[index] for index in indexes], dim=0)

    # ------------------------------------------------------------------ decoder backend + the device-side conversion
    def _next_decode(self) -> Any:
        return self._dec.decode_next()

    def _seek_and_decode(self, index: int) -> Any:
        return self._dec.seek_and_decode(index)

    def _convert(self, surface: Any, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """pitched surface -> (H, W) uint8 tensor (or into `out`, e.g. a slot of StreamPredictor's frame ring)"""
        ptr, pitch, keep = surface
        dev = self._device()
        if out is None:
            out = torch.empty(self.height, self.width, dtype=torch.uint8, device=dev)
        lib = self._lib if self._lib is not None else cabi.load()
        args = cabi.make("mds_frame_luma_args", width=self.width, height=self.height, pitch=int(pitch), count=1, src=int(ptr),
                         surface_stride=0, dst=out)
        stream = torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0
        lib.check(lib.fn["frame_luma"](C.byref(args), stream), "frame_luma")
        self._keep = (keep, out)            # the copy is asynchronous: the surface must outlive it
        return out
