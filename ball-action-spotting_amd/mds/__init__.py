"""mds — MI355X-native MultiDimStacker hot path (drop-in for src/models/multidim_stacker.py)."""
from .module import MultiDimStacker  # noqa: F401
from .cabi import MdsError, load  # noqa: F401
from . import train  # noqa: F401  (FocalLoss / FusedAdamW / ModelEma: SURVEY 8(f) N2)
from . import augment  # noqa: F401  (TrainAugmentations / get_train_augmentations: SURVEY 8(f) N3)
from . import frames  # noqa: F401  (RocDecFrameFetcher: SURVEY 8(f) N4, the device side)
