"""TEST INFRASTRUCTURE — CPU restatement of the reference's GPU augmentation pipeline.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this file; the product (mds.augment) never does.

Restates, stage by stage and in the reference's order, what `get_train_augmentations((1280, 736))` applies to a (B, T, H, W)
float batch in [0, 1] under no_grad (src/ball_action/augmentations.py:7-22, call site src/argus_models.py:49-53):

    RandomCameraMove((-2.5, 2.5), (0.1, 0.05), (0.95, 1.05), p=0.2)      src/augmentations.py:42-78  (the reference's own code)
    kornia.augmentation.RandomRotation(degrees=(-2.5, 2.5), p=0.3)
    RandomResizedCrop(size, scale=(0.9, 1.0), ratio=(r - 0.1, r + 0.1), p=0.8)      r = H / W
    RandomHorizontalFlip(p=0.5)
    RandomSharpness(sharpness=1., p=0.2)
    RandomMotionBlur(kernel_size=11, angle=7.5, direction=1.0, p=0.2)
    RandomBrightness(brightness=(0.8, 1.2), p=0.3)
    RandomContrast(contrast=(0.8, 1.2), p=0.3)
    RandomPosterize(bits=3, p=0.2)
    RandomGaussianNoise(mean=0., std=0.05, p=0.2)

PARITY UNPINNED for the kornia parts: kornia==0.6.12 (requirements.txt:10) is neither under /root/reference nor in this image, so
its functions (get_rotation_matrix2d / get_affine_matrix2d / warp_affine / rotate / crop_by_indices+resize / sharpness /
get_motion_kernel2d+filter2d / adjust_brightness (additive) / adjust_contrast (multiplicative) / posterize / the parameter
generators) are restated from their published algorithm [memory]; every such function says so.  What IS pinned: the reference's
own file src/augmentations.py (tensor_linspace, the per-frame interpolation of two affine parameter sets, `.T` layouts), which
tests/golden/make_golden.py executes with these kornia restatements seeded in as placeholder modules (augment_camera_move.npz).

A kornia 2D augmentation treats the T frames of a window as the channels of one image: ONE parameter set per sample for all
frames (RandomCameraMove alone varies per frame), applied to a sample with probability p.
"""
from __future__ import annotations

import math
import random

import torch
import torch.nn.functional as F

TRAIN_PIPELINE = dict(camera=dict(degrees=(-2.5, 2.5), translate=(0.1, 0.05), scale=(0.95, 1.05), p=0.2),
                      rotation=dict(degrees=(-2.5, 2.5), p=0.3),
                      crop=dict(scale=(0.9, 1.0), dratio=0.1, p=0.8),
                      flip=dict(p=0.5),
                      sharpness=dict(sharpness=1.0, p=0.2),
                      motion_blur=dict(kernel_size=11, angle=7.5, direction=1.0, p=0.2),
                      brightness=dict(brightness=(0.8, 1.2), p=0.3),
                      contrast=dict(contrast=(0.8, 1.2), p=0.3),
                      posterize=dict(bits=3, p=0.2),
                      noise=dict(mean=0.0, std=0.05, p=0.2))


# ---------------------------------------------------------------------------------------------- kornia geometry [memory]
def get_rotation_matrix2d(center, angle, scale):
    """kornia.geometry.transform.get_rotation_matrix2d: center (B,2) = (x, y), angle (B,) degrees (positive = counter-clockwise
    in image coordinates, the OpenCV convention), scale (B,2) -> (B,2,3)"""
    a = torch.deg2rad(angle)
    c, s_ = torch.cos(a), torch.sin(a)
    rot = torch.stack([c, s_, -s_, c], -1).view(-1, 2, 2)
    rot = rot @ torch.diag_embed(scale)
    alpha, beta = rot[:, 0, 0], rot[:, 0, 1]
    x, y = center[:, 0], center[:, 1]
    m = torch.zeros(center.shape[0], 2, 3, dtype=center.dtype)
    m[:, :2, :2] = rot
    m[:, 0, 2] = (1 - alpha) * x - beta * y
    m[:, 1, 2] = beta * x + (1 - alpha) * y
    return m


def get_affine_matrix2d(translations, center, scale, angle):
    """kornia.geometry.transform.get_affine_matrix2d without shear: rotation by -angle about `center`, scale, then translation;
    (B,3,3)"""
    m = get_rotation_matrix2d(center, -angle, scale)
    m[..., 2] += translations
    out = torch.zeros(m.shape[0], 3, 3, dtype=m.dtype)
    out[:, :2] = m
    out[:, 2, 2] = 1.0
    return out


def warp_affine(src, m, dsize, mode="bilinear"):
    """kornia.geometry.transform.warp_affine(src (B,C,H,W), M (B,2,3) mapping SOURCE pixels to DESTINATION pixels, dsize,
    mode, padding_mode='zeros', align_corners=True): dst(p) = src(M^-1 p), zeros outside"""
    b, c, h, w = src.shape
    ho, wo = dsize
    m3 = torch.zeros(b, 3, 3, dtype=torch.float64)
    m3[:, :2] = m.double()
    m3[:, 2, 2] = 1.0
    inv = torch.linalg.inv(m3)

    def norm(hh, ww):      # pixel -> [-1, 1] (align_corners=True)
        return torch.tensor([[2.0 / max(ww - 1, 1e-14), 0, -1.0], [0, 2.0 / max(hh - 1, 1e-14), -1.0], [0, 0, 1.0]], dtype=torch.float64)
    theta = norm(h, w) @ inv @ torch.linalg.inv(norm(ho, wo))            # dst normalised -> src normalised
    grid = F.affine_grid(theta[:, :2].to(src.dtype), [b, c, ho, wo], align_corners=True)
    return F.grid_sample(src, grid, mode=mode, padding_mode="zeros", align_corners=True)


def image_center(h, w, n=1, dtype=torch.float32):
    """kornia _compute_tensor_center: ((W - 1) / 2, (H - 1) / 2)"""
    return torch.tensor([[(w - 1) / 2.0, (h - 1) / 2.0]], dtype=dtype).repeat(n, 1)


# ---------------------------------------------------------------------------------------------- stages
def tensor_linspace(start, end, steps):
    """src/augmentations.py:11-39 (value-identical)"""
    w_e = torch.linspace(0, 1, steps=steps).to(start)
    w_s = torch.linspace(1, 0, steps=steps).to(start)
    return w_s * start.unsqueeze(-1) + w_e * end.unsqueeze(-1)


def camera_move(frames, params):
    """RandomCameraMove.forward for ONE selected sample (src/augmentations.py:62-77): frames (T,H,W); params = two affine
    parameter sets {"translations" (2,2), "center" (2,2), "scale" (2,2), "angle" (2,)} interpolated over the T frames"""
    t, h, w = frames.shape
    tr = tensor_linspace(params["translations"][0], params["translations"][1], t).T
    ce = tensor_linspace(params["center"][0], params["center"][1], t).T
    sc = tensor_linspace(params["scale"][0], params["scale"][1], t).T
    an = tensor_linspace(params["angle"][0], params["angle"][1], t)
    m = get_affine_matrix2d(tr.float(), ce.float(), sc.float(), an.float())
    return warp_affine(frames[:, None], m[:, :2, :], (h, w)).squeeze(1)


def camera_move_matrices(params, t):
    """the T per-frame 3x3 matrices (source -> destination pixels) camera_move warps with"""
    tr = tensor_linspace(params["translations"][0], params["translations"][1], t).T
    ce = tensor_linspace(params["center"][0], params["center"][1], t).T
    sc = tensor_linspace(params["scale"][0], params["scale"][1], t).T
    an = tensor_linspace(params["angle"][0], params["angle"][1], t)
    return get_affine_matrix2d(tr.float(), ce.float(), sc.float(), an.float())


def rotation(frames, angle):
    """kornia RandomRotation.apply_transform -> rotate / affine: about the image centre, bilinear, zeros, align_corners=True"""
    t, h, w = frames.shape
    m = get_rotation_matrix2d(image_center(h, w), torch.tensor([float(angle)]), torch.ones(1, 2))
    return warp_affine(frames[None], m, (h, w))[0]


def resized_crop(frames, box):
    """kornia RandomResizedCrop(cropping_mode='slice'): crop_by_indices + resize(bilinear, align_corners=True); box = (x, y, w, h)"""
    t, h, w = frames.shape
    x0, y0, cw, ch = box
    return F.interpolate(frames[None, :, y0:y0 + ch, x0:x0 + cw], size=(h, w), mode="bilinear", align_corners=True)[0]


def hflip(frames):
    return torch.flip(frames, dims=[-1])


def sharpness(frames, factor):
    """kornia.enhance.sharpness: 3x3 [[1,1,1],[1,5,1],[1,1,1]]/13 smoothing of the interior (border pixels keep their value), clamped,
    blended  out = blurred + (input - blurred) * factor   (factor 1 = input, 0 = blurred; clamped only outside (0, 1))"""
    t = frames.shape[0]
    k = torch.tensor([[1.0, 1, 1], [1, 5, 1], [1, 1, 1]]) / 13.0
    deg = F.conv2d(frames[None], k.view(1, 1, 3, 3).repeat(t, 1, 1, 1), groups=t)[0].clamp(0.0, 1.0)
    res = frames.clone()
    res[:, 1:-1, 1:-1] = deg
    if factor == 0:
        return res
    if factor == 1:
        return frames.clone()
    out = res + (frames - res) * factor
    return out if 0 < factor < 1 else out.clamp(0.0, 1.0)


def motion_kernel(ksize, angle, direction):
    """kornia get_motion_kernel2d(ksize, angle, direction, mode='nearest'): a horizontal line whose weights go linearly from d to
    1 - d (d = (direction + 1) / 2), rotated by `angle` with nearest sampling, normalised to sum 1; (ksize, ksize)"""
    d = (min(max(float(direction), -1.0), 1.0) + 1.0) / 2.0
    k = torch.tensor([d + ((1 - 2 * d) / (ksize - 1)) * i for i in range(ksize)])
    ker = torch.zeros(ksize, ksize)
    ker[ksize // 2] = k
    m = get_rotation_matrix2d(image_center(ksize, ksize), torch.tensor([float(angle)]), torch.ones(1, 2))
    ker = warp_affine(ker[None, None], m, (ksize, ksize), mode="nearest")[0, 0]
    return ker / ker.sum()


def motion_blur(frames, kernel):
    """kornia filter2d(input, kernel, border_type='constant'): cross-correlation with zero padding, one kernel for all frames"""
    t = frames.shape[0]
    ks = kernel.shape[-1]
    return F.conv2d(F.pad(frames[None], [ks // 2] * 4), kernel.view(1, 1, ks, ks).repeat(t, 1, 1, 1), groups=t)[0]


def brightness(frames, factor):
    """kornia RandomBrightness -> adjust_brightness(input, factor - 1): ADDITIVE, clamped to [0, 1]"""
    return (frames + (factor - 1.0)).clamp(0.0, 1.0)


def contrast(frames, factor):
    """kornia RandomContrast -> adjust_contrast(input, factor): MULTIPLICATIVE, clamped to [0, 1]"""
    return (frames * factor).clamp(0.0, 1.0)


def posterize(frames, bits):
    """kornia.enhance.posterize: keep the `bits` most significant bits of the 8-bit value (bits 8 = identity, 0 = zeros)"""
    bits = int(bits)
    if bits >= 8:
        return frames.clone()
    if bits <= 0:
        return torch.zeros_like(frames)
    x = (frames * 255.0).to(torch.uint8)
    shift = 8 - bits
    return (((x >> shift) << shift).float()) / 255.0


def gaussian_noise(frames, noise, std, mean=0.0):
    """kornia RandomGaussianNoise: input + randn_like(input) * std + mean (no clamp); `noise` is the standard-normal draw"""
    return frames + noise * std + mean


# ---------------------------------------------------------------------------------------------- parameter generators [memory]
def sample_params(b, t, h, w, gen: torch.Generator, cfg=TRAIN_PIPELINE):
    """one dict per sample with the stages that fire and their parameters (distributions of kornia 0.6.12's generators)"""
    def u(lo, hi, n=None):
        r = torch.rand(n or 1, generator=gen) * (hi - lo) + lo
        return r if n else r.item()

    def fire(p):
        return torch.rand(1, generator=gen).item() < p
    out = []
    for _ in range(b):
        s = {}
        c = cfg["camera"]
        if fire(c["p"]):     # src/augmentations.py:57: `random.random() > p: continue`; AffineGenerator on a batch of 2
            s["camera"] = dict(angle=u(*c["degrees"], 2),
                               translations=torch.stack([u(-c["translate"][0] * w, c["translate"][0] * w, 2),
                                                         u(-c["translate"][1] * h, c["translate"][1] * h, 2)], -1),
                               center=image_center(h, w, 2),
                               scale=u(*c["scale"], 2)[:, None].repeat(1, 2))
        if fire(cfg["rotation"]["p"]):
            s["rotation"] = u(*cfg["rotation"]["degrees"])
        c = cfg["crop"]
        if fire(c["p"]):
            r0 = h / w
            lo, hi = r0 - c["dratio"], r0 + c["dratio"]
            box = None
            for _try in range(10):      # ResizedCropGenerator: 10 attempts, the first valid one wins
                area = u(*c["scale"]) * h * w
                ar = math.exp(u(math.log(lo), math.log(hi)))
                ch, cw = math.floor(round(math.sqrt(area * ar))), math.floor(round(math.sqrt(area / ar)))   # kornia swaps: ratio is h / w
                if 0 < ch < h and 0 < cw < w and box is None:
                    box = (cw, ch)
            if box is None:
                box = (w, h)
            cw, ch = box
            s["crop"] = (int(math.floor(u(0, w - cw + 1))), int(math.floor(u(0, h - ch + 1))), cw, ch)
        if fire(cfg["flip"]["p"]):
            s["flip"] = True
        if fire(cfg["sharpness"]["p"]):
            s["sharpness"] = u(0.0, cfg["sharpness"]["sharpness"])
        c = cfg["motion_blur"]
        if fire(c["p"]):
            s["motion_blur"] = dict(ksize=c["kernel_size"], angle=u(-c["angle"], c["angle"]), direction=u(-c["direction"], c["direction"]))
        if fire(cfg["brightness"]["p"]):
            s["brightness"] = u(*cfg["brightness"]["brightness"])
        if fire(cfg["contrast"]["p"]):
            s["contrast"] = u(*cfg["contrast"]["contrast"])
        if fire(cfg["posterize"]["p"]):
            s["posterize"] = int(u(cfg["posterize"]["bits"], 8))
        if fire(cfg["noise"]["p"]):
            s["noise"] = dict(std=cfg["noise"]["std"], mean=cfg["noise"]["mean"], seed=int(torch.randint(0, 2**31 - 1, (1,), generator=gen)))
        out.append(s)
    return out


# ---------------------------------------------------------------------------------------------- the pipeline
def apply_reference_order(x, params, noise=None):
    """the reference pipeline as written: every stage resamples / filters the previous stage's OUTPUT (three bilinear
    resamplings in a row when camera move, rotation and crop all fire).  x (B,T,H,W); noise: optional (B,T,H,W) standard normal"""
    out = x.clone()
    for i, s in enumerate(params):
        f = out[i]
        if "camera" in s:
            f = camera_move(f, s["camera"])
        if "rotation" in s:
            f = rotation(f, s["rotation"])
        if "crop" in s:
            f = resized_crop(f, s["crop"])
        if "flip" in s:
            f = hflip(f)
        f = photometric(f, s, None if noise is None else noise[i])
        out[i] = f
    return out


def photometric(f, s, noise):
    if "sharpness" in s:
        f = sharpness(f, s["sharpness"])
    if "motion_blur" in s:
        mb = s["motion_blur"]
        f = motion_blur(f, motion_kernel(mb["ksize"], mb["angle"], mb["direction"]))
    if "brightness" in s:
        f = brightness(f, s["brightness"])
    if "contrast" in s:
        f = contrast(f, s["contrast"])
    if "posterize" in s:
        f = posterize(f, s["posterize"])
    if "noise" in s:
        f = gaussian_noise(f, noise, s["noise"]["std"], s["noise"]["mean"])
    return f


def geometric_matrices(s, t, h, w):
    """(T,3,3) float64: SOURCE -> DESTINATION pixel maps of the geometric stages of one sample composed in the reference's order
    (camera move, rotation, resized crop, flip); identity where a stage does not fire"""
    m = torch.eye(3, dtype=torch.float64).repeat(t, 1, 1)
    if "camera" in s:
        m = camera_move_matrices(s["camera"], t).double()
    if "rotation" in s:
        r = torch.eye(3, dtype=torch.float64)
        r[:2] = get_rotation_matrix2d(image_center(h, w, dtype=torch.float64), torch.tensor([float(s["rotation"])], dtype=torch.float64),
                                      torch.ones(1, 2, dtype=torch.float64))[0]
        m = r @ m
    if "crop" in s:
        x0, y0, cw, ch = s["crop"]
        sx, sy = (w - 1) / max(cw - 1, 1), (h - 1) / max(ch - 1, 1)
        c = torch.tensor([[sx, 0, -x0 * sx], [0, sy, -y0 * sy], [0, 0, 1]], dtype=torch.float64)
        m = c @ m
    if "flip" in s:
        fl = torch.tensor([[-1.0, 0, w - 1], [0, 1, 0], [0, 0, 1]], dtype=torch.float64)
        m = fl @ m
    return m


def apply_single_resampling(x, params, noise=None):
    """what the fused HIP pass computes: the geometric stages composed into ONE per-frame affine map and ONE bilinear resampling
    of the input (zeros outside), then the photometric stages as in the reference.  Equal to `apply_reference_order` whenever at
    most one resampling stage fires for a sample; with several it avoids the reference's repeated interpolation blur (and its
    intermediate zero borders) - a deliberate, documented difference of the fused kernel."""
    out = x.clone()
    b, t, h, w = x.shape
    for i, s in enumerate(params):
        f = out[i]
        if any(k in s for k in ("camera", "rotation", "crop", "flip")):
            m = geometric_matrices(s, t, h, w)
            f = warp_affine(f[:, None], m[:, :2].float(), (h, w)).squeeze(1)
        out[i] = photometric(f, s, None if noise is None else noise[i])
    return out
