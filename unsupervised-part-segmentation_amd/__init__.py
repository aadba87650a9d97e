"""MI355X-native training path for the part-discovery model of
CompVis/unsupervised-part-segmentation (package directory ``unsupervised-part-segmentation_amd``,
importable as ``upsparts_amd`` through the alias module at the repository root).

``csrc/``   hand-written HIP kernels (gfx950) behind the C ABI of include/upsparts_hip.h
``lib``     ctypes binding (no CPU fallback)
``ops``     tap geometry + autograd wrappers
``nets``    encoder / mask decoder / hourglass / critic / perceptual-trunk builders
``model``   TrainModel + Trainer (the edflow surface of the reference)
``runner``  ``edflow -t config.yaml`` work-alike, synthetic dataset
"""
from .lib import UpsError, load, LIB_PATH  # noqa: F401

__all__ = ["UpsError", "load", "LIB_PATH"]
