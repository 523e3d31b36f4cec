"""MI355X-native (gfx950) hot path of the joint segmentation + disparity network of
cuevhv/PMT_learning_for_semantic_segmentation_and_disparity.

Importing the package loads libsdhip.so (hand-written HIP kernels behind a C ABI,
see include/sdhip.h).  There is no CPU / eager fallback.
"""
from . import _lib  # noqa: F401  (fails loudly when the HIP library is absent)
from ._lib import SdhipError, abi_version  # noqa: F401
