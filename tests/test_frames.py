"""SURVEY 8(f) N4, the part on this side of the decoder: mds.frames.RocDecFrameFetcher (the reference fetcher's interface and
behaviour, src/frame_fetchers/abstract.py + nvdec.py) over a SYNTHETIC decoder backend that hands out pitched NV12 surfaces, and
mds_frame_luma (pitched luma plane -> contiguous frames) against plain slicing."""
import pytest
import torch

from backends import be  # noqa: F401
from mds import cabi
from mds.frames import RocDecFrameFetcher, open_rocdecode


class FakeDecoder:
    """NV12 surfaces (luma rows then interleaved chroma rows) with a row pitch larger than the width; frame i's luma is a known
    pattern.  Frame 7 is 'corrupt': decoding it raises, like a broken packet."""

    def __init__(self, device, n=12, w=50, h=20, pitch=64):
        self.num_frames, self.width, self.height, self.pitch, self.device = n, w, h, pitch, device
        g = torch.Generator().manual_seed(3)
        self.luma = torch.randint(0, 256, (n, h, w), generator=g, dtype=torch.uint8)
        self.pos = 0
        self.decoded = []

    def _surface(self, i):
        if i == 7:
            raise IOError("corrupt packet")
        buf = torch.full((self.height * 3 // 2, self.pitch), 99, dtype=torch.uint8)      # padding / chroma bytes must never show up
        buf[:self.height, :self.width] = self.luma[i]
        buf = buf.to(self.device)
        self.decoded.append(i)
        return (buf.data_ptr(), self.pitch, buf)

    def decode_next(self):
        i = self.pos
        self.pos += 1
        return self._surface(i)

    def seek_and_decode(self, index):
        self.pos = index + 1
        return self._surface(index)


def _fetcher(be):
    dec = FakeDecoder(be.device)
    f = RocDecFrameFetcher("video.mkv", 0, decoder=dec)
    f._lib = be.lib if be.name == "emu" else None
    return f, dec


def test_fetch_frame_sequential_seek_and_error_replacement(be):
    f, dec = _fetcher(be)
    assert (f.num_frames, f.width, f.height, f.current_index) == (12, 50, 20, -1)
    a = f.fetch_frame()                       # sequential: frame 0
    b = f.fetch_frame()                       # frame 1
    c = f.fetch_frame(5)                      # seek
    d = f.fetch_frame()                       # continues after the seek: frame 6
    e = f.fetch_frame()                       # frame 7 is corrupt -> logged, zero frame (the decode raised BEFORE `_current_index += 1`, abstract.py:30-31)
    g = f.fetch_frame()                       # frame 8
    be.sync()
    for t, i in ((a, 0), (b, 1), (c, 5), (d, 6), (g, 8)):
        assert t.dtype == torch.uint8 and t.shape == (20, 50) and torch.equal(t.cpu(), dec.luma[i]), i
    assert e.abs().sum().item() == 0 and f.current_index == 7     # the reference's bookkeeping: one behind after a failed sequential decode
    assert f.fetch_frame(12).abs().sum().item() == 0          # out of range: error path, zero frame (abstract.py:33-48)
    f.fetch_frame(11)
    assert f.fetch_frame().abs().sum().item() == 0            # past the end of the stream


def test_fetch_frames_sweeps_forward_once(be):
    f, dec = _fetcher(be)
    idx = [2, 4, 6]                                           # a stack with frame_stack_step 2 (src/indexes.py)
    clip = f.fetch_frames(idx)
    be.sync()
    assert clip.shape == (3, 20, 50) and torch.equal(clip.cpu(), dec.luma[idx])
    assert dec.decoded == [2, 3, 4, 5, 6]                     # one seek, then sequential decode; skipped frames decoded and dropped


def test_fetch_into_writes_in_place_any_order_with_repeats(be):
    """the clip is written in its final place (no per-frame tensors, no stack): rows follow the ORDER of `indexes`, a repeated
    index is decoded once, a corrupt frame is a zero row, and the destination may be a view of a larger ring"""
    f, dec = _fetcher(be)
    ring = torch.full((6, 20, 50), 7, dtype=torch.uint8, device=be.device)
    idx = [8, 6, 7, 6]
    out = f.fetch_into(idx, ring[1:5])
    be.sync()
    assert out.data_ptr() == ring[1:5].data_ptr()
    assert torch.equal(ring[1].cpu(), dec.luma[8]) and torch.equal(ring[2].cpu(), dec.luma[6]) and torch.equal(ring[4].cpu(), dec.luma[6])
    assert ring[3].abs().sum().item() == 0                      # frame 7 is corrupt
    assert (ring[0] == 7).all() and (ring[5] == 7).all()        # nothing outside the destination was touched
    assert dec.decoded == [6, 8]                                # one seek + sequential decode (7 raised inside the decoder)


@pytest.mark.parametrize("w,h,pitch,count", [(50, 20, 64, 1), (1280, 720, 1280, 2), (720, 33, 768, 3), (17, 5, 32, 2)])
def test_frame_luma_kernel(be, w, h, pitch, count):
    g = torch.Generator().manual_seed(w + h)
    surf = torch.randint(0, 256, (count, h * 3 // 2 + 1, pitch), generator=g, dtype=torch.uint8)
    sd = be.t(surf)
    dst = torch.full((count, h, w), 7, dtype=torch.uint8, device=be.device)
    be.call("frame_luma", cabi.make("mds_frame_luma_args", width=w, height=h, pitch=pitch, count=count, src=sd,
                                    surface_stride=surf.stride(0), dst=dst))
    be.sync()
    assert torch.equal(dst.cpu(), surf[:, :h, :w])


@pytest.mark.parametrize("row_bytes,src_pitch,dst_pitch,nrows", [(5220, 5220, 5220, 3), (4096, 8192, 4096, 5), (1000, 1024, 2048, 7),
                                                                 (921600, 921600, 921600, 2), (48, 64, 48, 320)])
def test_copy_rows_kernel(be, row_bytes, src_pitch, dst_pitch, nrows):
    """mds_copy_rows (the predictor's ring / store addressing): dst row dst_slot[r] <- src row src_slot[r], slot numbers in the
    kernel arguments; 16-byte vectors where the rows allow, bytes otherwise; bytes between rows and untouched slots stay"""
    g = torch.Generator().manual_seed(row_bytes + nrows)
    nsrc, ndst = nrows + 3, nrows + 2
    src = torch.randint(0, 256, (nsrc, src_pitch), generator=g, dtype=torch.uint8)
    dst0 = torch.randint(0, 256, (ndst, dst_pitch), generator=g, dtype=torch.uint8)
    src_slot = torch.randint(0, nsrc, (nrows,), generator=g).tolist()
    dst_slot = torch.randperm(ndst, generator=g)[:nrows].tolist()
    sd, dd = be.t(src), be.t(dst0.clone())
    be.call("copy_rows", cabi.make("mds_copy_rows_args", dst=dd, src=sd, dst_pitch=dst_pitch, src_pitch=src_pitch, row_bytes=row_bytes,
                                   nrows=nrows, dst_slot=dst_slot, src_slot=src_slot))
    be.sync()
    want = dst0.clone()
    for r in range(nrows):
        want[dst_slot[r], :row_bytes] = src[src_slot[r], :row_bytes]
    assert torch.equal(dd.cpu(), want)


def test_copy_rows_refuses_more_rows_than_its_argument_block_holds(be):
    t = be.t(torch.zeros(4, 16, dtype=torch.uint8))
    a = cabi.make("mds_copy_rows_args", dst=t, src=t, dst_pitch=16, src_pitch=16, row_bytes=16, nrows=cabi.MDS_COPY_ROWS_MAX + 1)
    assert be.lib.fn["copy_rows"](a, 0) != 0


def test_rocdecode_backend_is_reported_missing_not_faked():
    with pytest.raises(RuntimeError, match="rocDecode"):
        open_rocdecode("video.mkv", 0)
