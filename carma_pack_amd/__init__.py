"""carma_pack_amd -- MI355X-native CARMA Kalman log-likelihood path behind carma_pack's API.

Importing this package loads ``libcarma_mi355.so`` (hand-written HIP for gfx950) through its
C ABI; it fails loudly if the library has not been built.
"""
from . import _lib  # noqa: F401  (raises ImportError when the HIP library is missing)
from ._lib import CarmaDeviceError, CarmaError, Context, kfilter_car1, kfilter_carma  # noqa: F401

__all__ = ["Context", "kfilter_carma", "kfilter_car1", "CarmaError", "CarmaDeviceError"]
