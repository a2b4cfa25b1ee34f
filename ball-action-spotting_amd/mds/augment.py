"""SURVEY §8(f) N3 — the reference's GPU training augmentations as fused HIP passes.

Drop-in for ``get_train_augmentations(size)`` of ``src/ball_action/augmentations.py:7-22`` (identical to
``src/action/augmentations.py:7-22``): an ``nn.Module`` the trainer calls on the (B, T, H, W) float batch under ``no_grad``
(``src/argus_models.py:49-53``).  The reference runs ten kornia stages one after the other - each reads and writes the whole
226 MB batch, three of them resample it bilinearly in a row.  Here

* the parameters of every stage are drawn on the host with the reference's distributions (kornia 0.6.12's generators and
  ``RandomCameraMove``'s two interpolated affine parameter sets, ``src/augmentations.py:55-72``);
* camera move, rotation, resized crop and horizontal flip are composed (float64, host) into ONE destination -> source
  affine map per frame, and the batch is resampled ONCE (``mds_aug_pass``, mode WARP) - no repeated interpolation blur, no
  intermediate zero borders; identical to the reference whenever at most one resampling stage fires for a sample;
* brightness, contrast, posterize and Gaussian noise are applied in the same pass before the only store;
* sharpness (3x3) and motion blur (11x11 line kernel, built on the host) are spatial filters of the previous stage's output:
  the samples that drew them (p = 0.2 each) take one extra pass each through a scratch buffer, the others do not.

No CPU fallback: CPU tensors raise unless the test-suite's kernel simulator has been injected (``TrainAugmentations._lib``).
"""
from __future__ import annotations

import ctypes as C
import math
import random
from typing import List, Optional

import numpy as np
import torch
from torch import nn

from . import cabi

MAX_TAPS = cabi.MDS_AUG_MAX_TAPS


# ------------------------------------------------------------------------------------------------ affine algebra (host, float64)
def _rotation_matrix(cx, cy, angle_deg, scale=1.0):
    """kornia get_rotation_matrix2d (OpenCV convention: positive angle = counter-clockwise), 3x3 source -> destination"""
    a = math.radians(angle_deg)
    al, be = scale * math.cos(a), scale * math.sin(a)
    return np.array([[al, be, (1 - al) * cx - be * cy], [-be, al, be * cx + (1 - al) * cy], [0, 0, 1.0]])


def _affine_matrix(tx, ty, cx, cy, scale, angle_deg):
    """kornia get_affine_matrix2d without shear: rotation by -angle about the centre, scale, then translation"""
    m = _rotation_matrix(cx, cy, -angle_deg, scale)
    m[0, 2] += tx
    m[1, 2] += ty
    return m


def motion_kernel(ksize: int, angle: float, direction: float) -> np.ndarray:
    """kornia get_motion_kernel2d(mode='nearest'): a horizontal line with weights going linearly from d to 1 - d
    (d = (direction + 1) / 2), rotated by `angle` about the kernel centre with nearest sampling, normalised to sum 1"""
    d = (min(max(direction, -1.0), 1.0) + 1.0) / 2.0
    line = (d + ((1 - 2 * d) / (ksize - 1)) * np.arange(ksize)).astype(np.float32)
    base = np.zeros((ksize, ksize), dtype=np.float32)
    base[ksize // 2] = line
    c = (ksize - 1) / 2.0
    inv = np.linalg.inv(_rotation_matrix(c, c, angle))
    ys, xs = np.mgrid[0:ksize, 0:ksize]
    sx = (inv[0, 0] * xs + inv[0, 1] * ys + inv[0, 2]).astype(np.float32)
    sy = (inv[1, 0] * xs + inv[1, 1] * ys + inv[1, 2]).astype(np.float32)
    ix, iy = np.rint(sx).astype(np.int64), np.rint(sy).astype(np.int64)      # grid_sample 'nearest' rounds half to even
    ok = (ix >= 0) & (ix < ksize) & (iy >= 0) & (iy < ksize)
    out = np.where(ok, base[np.clip(iy, 0, ksize - 1), np.clip(ix, 0, ksize - 1)], 0.0).astype(np.float32)
    return out / out.sum()


class TrainAugmentations(nn.Module):
    """``get_train_augmentations(size)`` with size = (width, height) like the reference's configs (image_size)."""

    def __init__(self, size, seed: Optional[int] = None, compose_geometric: bool = False):
        """compose_geometric=False (default): the REFERENCE ORDER - each geometric stage resamples the previous stage's output
        as kornia's nn.Sequential does (src/ball_action/augmentations.py:10-13); result-identical to the reference for every
        sample (0.46 ms per 4x15x736x1280 batch).  True: the opt-in fast path - camera move, rotation, resized crop and flip
        composed into ONE resampling of the input (0.36 ms; less interpolation blur and no intermediate zero borders than the
        reference, identical to it only when at most one of them fires: 64 % of the samples at the reference's probabilities)."""
        super().__init__()
        self.compose_geometric = bool(compose_geometric)
        self.width, self.height = int(size[0]), int(size[1])
        r = self.height / self.width                       # `size = size[::-1]; ratio = size[0] / size[1]`
        self.cfg = dict(camera=dict(degrees=(-2.5, 2.5), translate=(0.1, 0.05), scale=(0.95, 1.05), p=0.2),
                        rotation=dict(degrees=(-2.5, 2.5), p=0.3),
                        crop=dict(scale=(0.9, 1.0), ratio=(r - 0.1, r + 0.1), p=0.8),
                        flip=dict(p=0.5), sharpness=dict(sharpness=1.0, p=0.2),
                        motion_blur=dict(kernel_size=11, angle=7.5, direction=1.0, p=0.2),
                        brightness=dict(brightness=(0.8, 1.2), p=0.3), contrast=dict(contrast=(0.8, 1.2), p=0.3),
                        posterize=dict(bits=3, p=0.2), noise=dict(mean=0.0, std=0.05, p=0.2))
        self.rng = random.Random(seed)
        self._lib = None          # tests inject the kernel simulator here
        self._scratch = {}

    # ------------------------------------------------------------------ parameter generation (reference distributions)
    def sample_params(self, b: int, t: int, h: int, w: int) -> List[dict]:
        u, cfg = self.rng.uniform, self.cfg
        out = []
        for _ in range(b):
            s = {}
            c = cfg["camera"]
            if not (self.rng.random() > c["p"]):           # src/augmentations.py:57
                # AffineGenerator on a batch of 2: start and end of the camera move (angle, translation, isotropic scale)
                s["camera"] = dict(angle=[u(*c["degrees"]) for _ in range(2)],
                                   translations=[[u(-c["translate"][0] * w, c["translate"][0] * w), u(-c["translate"][1] * h, c["translate"][1] * h)]
                                                 for _ in range(2)],
                                   center=[[(w - 1) / 2.0, (h - 1) / 2.0]] * 2,
                                   scale=[[sc, sc] for sc in (u(*c["scale"]), u(*c["scale"]))])
            if self.rng.random() < cfg["rotation"]["p"]:
                s["rotation"] = u(*cfg["rotation"]["degrees"])
            c = cfg["crop"]
            if self.rng.random() < c["p"]:
                box = None
                for _try in range(10):                     # ResizedCropGenerator: ten attempts, the first valid one is used
                    area = u(*c["scale"]) * h * w
                    ar = math.exp(u(math.log(c["ratio"][0]), math.log(c["ratio"][1])))
                    ch, cw = math.floor(round(math.sqrt(area * ar))), math.floor(round(math.sqrt(area / ar)))    # kornia: ratio = h / w
                    if box is None and 0 < ch < h and 0 < cw < w:
                        box = (cw, ch)
                cw, ch = box or (w, h)
                s["crop"] = (int(math.floor(u(0, w - cw + 1))), int(math.floor(u(0, h - ch + 1))), cw, ch)
            if self.rng.random() < cfg["flip"]["p"]:
                s["flip"] = True
            if self.rng.random() < cfg["sharpness"]["p"]:
                s["sharpness"] = u(0.0, cfg["sharpness"]["sharpness"])
            c = cfg["motion_blur"]
            if self.rng.random() < c["p"]:
                s["motion_blur"] = dict(ksize=c["kernel_size"], angle=u(-c["angle"], c["angle"]), direction=u(-c["direction"], c["direction"]))
            if self.rng.random() < cfg["brightness"]["p"]:
                s["brightness"] = u(*cfg["brightness"]["brightness"])
            if self.rng.random() < cfg["contrast"]["p"]:
                s["contrast"] = u(*cfg["contrast"]["contrast"])
            if self.rng.random() < cfg["posterize"]["p"]:
                s["posterize"] = int(u(cfg["posterize"]["bits"], 8))
            if self.rng.random() < cfg["noise"]["p"]:
                s["noise"] = dict(std=cfg["noise"]["std"], mean=cfg["noise"]["mean"], seed=self.rng.randrange(2 ** 31 - 1))
            out.append(s)
        return out

    # ------------------------------------------------------------------ host tables
    @staticmethod
    def stage_matrices(s: dict, t: int, h: int, w: int) -> List[np.ndarray]:
        """source -> destination matrices ((T, 3, 3) float64) of the geometric stages of one sample that fire, in the
        reference's order: camera move, rotation, resized crop; the horizontal flip - a pure re-indexing of the destination,
        exact under bilinear sampling - is folded into the last of them (or stands alone)"""
        out = []
        if "camera" in s:
            p = s["camera"]
            fwd = np.tile(np.eye(3), (t, 1, 1))

            def lin(a, b):        # src/augmentations.py tensor_linspace: start * linspace(1, 0) + end * linspace(0, 1)
                we = np.linspace(0.0, 1.0, t, dtype=np.float32).astype(np.float64)
                ws = np.linspace(1.0, 0.0, t, dtype=np.float32).astype(np.float64)
                return ws * float(a) + we * float(b)
            tr = np.asarray(p["translations"], dtype=np.float64); ce = np.asarray(p["center"], dtype=np.float64)
            sc = np.asarray(p["scale"], dtype=np.float64); an = np.asarray(p["angle"], dtype=np.float64)
            tx, ty = lin(tr[0, 0], tr[1, 0]), lin(tr[0, 1], tr[1, 1])
            cx, cy = lin(ce[0, 0], ce[1, 0]), lin(ce[0, 1], ce[1, 1])
            ss, aa = lin(sc[0, 0], sc[1, 0]), np.radians(-lin(an[0], an[1]))        # get_affine_matrix2d rotates by -angle
            al, be = ss * np.cos(aa), ss * np.sin(aa)
            fwd[:, 0, 0], fwd[:, 0, 1], fwd[:, 0, 2] = al, be, (1 - al) * cx - be * cy + tx
            fwd[:, 1, 0], fwd[:, 1, 1], fwd[:, 1, 2] = -be, al, be * cx + (1 - al) * cy + ty
            out.append(fwd)
        if "rotation" in s:
            out.append(np.tile(_rotation_matrix((w - 1) / 2.0, (h - 1) / 2.0, float(s["rotation"])), (t, 1, 1)))
        if "crop" in s:
            x0, y0, cw, ch = s["crop"]
            sx, sy = (w - 1) / max(cw - 1, 1), (h - 1) / max(ch - 1, 1)         # resize(..., align_corners=True) of the slice
            out.append(np.tile(np.array([[sx, 0, -x0 * sx], [0, sy, -y0 * sy], [0, 0, 1.0]]), (t, 1, 1)))
        if "flip" in s:
            fl = np.array([[-1.0, 0, w - 1], [0, 1, 0], [0, 0, 1]])
            if out:
                out[-1] = fl @ out[-1]
            else:
                out.append(np.tile(fl, (t, 1, 1)))
        return out

    @staticmethod
    def frame_maps(s: dict, t: int, h: int, w: int, compose: bool = True) -> List[np.ndarray]:
        """(T, 6) float32 destination -> source maps, one per resampling pass of this sample: a single composed map
        (compose=True), or one per geometric stage in the reference's order; [] when no geometric stage fires"""
        ms = TrainAugmentations.stage_matrices(s, t, h, w)
        if not ms:
            return []
        if compose:
            fwd = ms[0]
            for m in ms[1:]:
                fwd = m @ fwd
            ms = [fwd]
        return [np.linalg.inv(m)[:, :2, :].reshape(t, 6).astype(np.float32) for m in ms]

    NWARP = 3          # resampling pass slots (camera move, rotation, resized crop in reference order; one when composed)
    SLOT_SHARP, SLOT_TAPS, NSLOTS = 3, 4, 5

    def _jobs(self, params, t, h, w):
        """per pass slot: the job table (ctypes array) of every sample; plus the (NWARP, B, T, 6) maps of the resampling slots"""
        Job = cabi.STRUCTS["mds_aug_job"]
        b = len(params)
        maps = np.zeros((self.NWARP, b, t, 6), dtype=np.float32)
        maps[..., 0] = 1.0
        maps[..., 4] = 1.0
        passes = [(Job * b)() for _ in range(self.NSLOTS)]
        used = [False] * self.NSLOTS
        need_scratch = 0
        for i, s in enumerate(params):
            mps = self.frame_maps(s, t, h, w, self.compose_geometric)
            chain = [("warp", j) for j in range(len(mps))] or [("copy", 0)]
            for j, mp in enumerate(mps):
                maps[j, i] = mp
            if "sharpness" in s:
                chain.append(("sharp", self.SLOT_SHARP))
            if "motion_blur" in s:
                chain.append(("taps", self.SLOT_TAPS))
            if len(chain) > 1 and chain[0][0] == "copy":
                chain = chain[1:]                         # no resampling: the first filter reads the input directly
            for k, (kind, slot) in enumerate(chain):
                jb = passes[slot][i]
                used[slot] = True
                jb.active = 1
                jb.src = 0 if k == 0 else 2 + (k - 1) % 2
                last = k == len(chain) - 1
                jb.dst = 1 if last else 2 + k % 2
                if not last:
                    need_scratch = max(need_scratch, 1 + k % 2)
                jb.mode = {"copy": cabi.MDS_AUG_COPY, "warp": cabi.MDS_AUG_WARP, "sharp": cabi.MDS_AUG_SHARP, "taps": cabi.MDS_AUG_TAPS}[kind]
                if kind == "sharp":
                    jb.sharp_factor = float(s["sharpness"])
                if kind == "taps":
                    mb = s["motion_blur"]
                    ker = motion_kernel(mb["ksize"], mb["angle"], mb["direction"])
                    ys, xs = np.nonzero(ker)
                    assert len(ys) <= MAX_TAPS, "motion kernel with more non-zero taps than MDS_AUG_MAX_TAPS"
                    jb.ntaps = len(ys)
                    for n_, (yy, xx) in enumerate(zip(ys, xs)):
                        jb.tap_dx[n_], jb.tap_dy[n_], jb.tap_w[n_] = int(xx) - mb["ksize"] // 2, int(yy) - mb["ksize"] // 2, float(ker[yy, xx])
                if last:
                    jb.point = 1
                    if "brightness" in s:
                        jb.bright_on, jb.bright_add = 1, float(s["brightness"]) - 1.0        # kornia: additive brightness
                    if "contrast" in s:
                        jb.contrast_on, jb.contrast_mul = 1, float(s["contrast"])
                    if "posterize" in s and int(s["posterize"]) < 8:
                        jb.posterize_bits = int(s["posterize"])
                    if "noise" in s:
                        jb.noise_on, jb.noise_std, jb.noise_mean = 1, float(s["noise"]["std"]), float(s["noise"]["mean"])
                        jb.noise_seed = int(s["noise"]["seed"]) & 0x7FFFFFFF
        return passes, used, maps, need_scratch

    # ------------------------------------------------------------------ forward
    def _library(self, x):
        if self._lib is not None:
            return self._lib
        if not x.is_cuda:
            raise cabi.MdsError("mds.augment runs on MI355X only: move the batch to cuda")
        return cabi.load()

    def prepare(self, params: List[dict], t: int, h: int, w: int, device) -> dict:
        """host side of one batch: job tables + maps in ONE device upload (parameters -> what the launches need)"""
        passes, used, maps, need_scratch = self._jobs(params, t, h, w)
        b = len(params)
        raw = b"".join(bytes(p) for p, u_ in zip(passes, used) if u_) + maps.tobytes()
        table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device, non_blocking=True)
        return dict(table=table, used=used, nused=sum(used), need_scratch=need_scratch, shape=(b, t, h, w), jsz=C.sizeof(cabi.STRUCTS["mds_aug_job"]) * b,
                    map_bytes=b * t * 6 * 4)

    @torch.no_grad()
    def forward(self, x: torch.Tensor, params: Optional[List[dict]] = None, noise: Optional[torch.Tensor] = None, prepared: Optional[dict] = None):
        """x: (B, T, H, W) float32 in [0, 1].  `params` (a list of per-sample dicts as `sample_params` returns) and `noise`
        (standard-normal draws, (B, T, H, W)) are injection points for parity tests; normally both are drawn here.
        `prepared` = the result of `prepare()` for this shape (bench: the launches alone)."""
        assert x.dim() == 4 and x.dtype == torch.float32, "augmentations take the (B, T, H, W) float32 frame batch"
        x = x.contiguous()
        b, t, h, w = x.shape
        lib = self._library(x)
        dev = x.device
        if prepared is None:
            if params is None:
                params = self.sample_params(b, t, h, w)
            prepared = self.prepare(params, t, h, w, dev)
            self.last_params = params
        assert prepared["shape"] == (b, t, h, w)
        table, used, jsz = prepared["table"], prepared["used"], prepared["jsz"]
        out = torch.empty_like(x)
        bufs = [x.data_ptr(), out.data_ptr(), 0, 0]
        for k in range(prepared["need_scratch"]):
            key = (k, x.shape, dev)
            if key not in self._scratch:
                self._scratch = {kk: v for kk, v in self._scratch.items() if kk[1:] == (x.shape, dev)}
                self._scratch[key] = torch.empty_like(x)
            bufs[2 + k] = self._scratch[key].data_ptr()
        maps_ptr = table.data_ptr() + prepared["nused"] * jsz
        stream = torch.cuda.current_stream(dev).cuda_stream if x.is_cuda else 0
        if noise is not None:
            noise = noise.to(device=dev, dtype=torch.float32).contiguous()
        with torch.cuda.device(dev) if x.is_cuda else _Null():
            k = 0
            for slot in range(self.NSLOTS):
                if not used[slot]:
                    continue
                mp = maps_ptr + (slot if slot < self.NWARP else 0) * prepared["map_bytes"]       # each resampling slot has its own maps
                args = cabi.make("mds_aug_args", B=b, T=t, H=h, W=w, buf=bufs, jobs=table.data_ptr() + k * jsz, maps=mp, noise=noise)
                lib.check(lib.fn["aug_pass"](C.byref(args), stream), "aug_pass")
                k += 1
        self._last = (table, x, out, noise)      # the launches are asynchronous: their operands outlive this call
        return out


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def get_train_augmentations(size, compose_geometric: bool = False) -> nn.Module:
    """src/ball_action/augmentations.py:7 / src/action/augmentations.py:7.  The default is the reference's stage order
    (result-identical to it); compose_geometric=True is the opt-in single-resampling form (see TrainAugmentations)."""
    return TrainAugmentations(size, compose_geometric=compose_geometric)
