"""Generate tests/golden/augment_camera_move.npz by running THE REFERENCE's own RandomCameraMove (build container only).

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_augment.py

/root/reference/src/augmentations.py is loaded by path.  Its kornia imports (kornia 0.6.12 is absent here) are satisfied by
placeholder modules whose content is this repo's own restatement (oracle/augment_ref.py: get_affine_matrix2d, warp_affine) and a
parameter generator that returns the two parameter sets stored in the fixture.  What the fixture pins is therefore the reference's
own file: tensor_linspace, the per-frame interpolation between two parameter sets, the `.T` layouts, the call conventions and the
`random.random() > p` selection - executed line for line.  Only inputs / parameters / expected outputs (data) are stored.
"""
import importlib.util
import os
import random
import sys
import types

sys.dont_write_bytecode = True
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import augment_ref as aug  # noqa: E402

REF_FILE = "/root/reference/src/augmentations.py"


class FixedAffineGenerator:
    """stands in for kornia.augmentation.random_generator.AffineGenerator: hands out the next stored parameter set"""
    queue = []

    def __init__(self, degrees, translate, scale):
        self.args = (degrees, translate, scale)

    def __call__(self, batch_shape):
        assert tuple(batch_shape)[:2] == (2, 1)
        return FixedAffineGenerator.queue.pop(0)


def load_reference():
    kornia = types.ModuleType("kornia")
    geometry = types.ModuleType("kornia.geometry")
    transform = types.ModuleType("kornia.geometry.transform")
    transform.get_affine_matrix2d = aug.get_affine_matrix2d
    transform.warp_affine = lambda src, m, dsize: aug.warp_affine(src, m, dsize)
    augmentation = types.ModuleType("kornia.augmentation")
    rg = types.ModuleType("kornia.augmentation.random_generator")
    rg.AffineGenerator = FixedAffineGenerator
    augmentation.random_generator = rg
    core = types.ModuleType("kornia.core")
    core.as_tensor = torch.as_tensor
    kornia.geometry, geometry.transform, kornia.augmentation, kornia.core = geometry, transform, augmentation, core
    for n, m in (("kornia", kornia), ("kornia.geometry", geometry), ("kornia.geometry.transform", transform),
                 ("kornia.augmentation", augmentation), ("kornia.augmentation.random_generator", rg), ("kornia.core", core)):
        sys.modules[n] = m
    spec = importlib.util.spec_from_file_location("ref_augmentations", REF_FILE)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ref = load_reference()
    g = torch.Generator().manual_seed(31)
    b, t, h, w = 3, 5, 12, 16
    x = torch.rand(b, t, h, w, generator=g)
    sets = []
    for _ in range(b):
        p = dict(angle=(torch.rand(2, generator=g) * 5 - 2.5), translations=torch.stack([(torch.rand(2, generator=g) * 2 - 1) * 0.1 * w,
                                                                                       (torch.rand(2, generator=g) * 2 - 1) * 0.05 * h], -1),
                 center=aug.image_center(h, w, 2), scale=(torch.rand(2, generator=g) * 0.1 + 0.95)[:, None].repeat(1, 2))
        sets.append(p)
    # p = 0.7 with a seeded `random`: the stored `selected` mask records which samples the reference moved
    random.seed(8)
    state = random.getstate()
    selected = [random.random() <= 0.7 for _ in range(b)]
    random.setstate(state)
    FixedAffineGenerator.queue = [s for s, sel in zip(sets, selected) if sel]
    mod = ref.RandomCameraMove((-2.5, 2.5), (0.1, 0.05), (0.95, 1.05), p=0.7)
    y = mod(x)
    assert not FixedAffineGenerator.queue
    lin = ref.tensor_linspace(torch.tensor([1.0, -2.0]), torch.tensor([3.0, 4.0]), 7)
    d = dict(x=x.numpy(), y=y.numpy(), selected=np.array(selected), linspace=lin.numpy())
    for i, s in enumerate(sets):
        for k, v in s.items():
            d[f"p{i}.{k}"] = v.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "augment_camera_move.npz"), **d)
    print({k: v.shape for k, v in d.items()}, "selected", selected)


if __name__ == "__main__":
    main()
