"""carma_pack_amd -- MI355X-native CARMA Kalman log-likelihood path behind carma_pack's API.

Importing this package loads ``libcarma_mi355.so`` (hand-written HIP for gfx950) through its
C ABI; it fails loudly if the library has not been built.

  carma_pack_amd._lib        ctypes binding of the C ABI (include/carma_mi355.h), ``Context``
  carma_pack_amd._carmcmc    drop-in for the reference's compiled extension ``carmcmc._carmcmc``
  carma_pack_amd.carma_pack  the reference's Python API (CarmaModel, CarmaSample, ...)
  carma_pack_amd.parallel    one-process-per-GPU sharding helpers (torch.distributed)
"""
from . import _lib  # noqa: F401  (raises ImportError when the HIP library is missing)
from ._lib import CarmaDeviceError, CarmaError, Context, kfilter_car1, kfilter_carma, kfilter_carma_batch  # noqa: F401
from .carma_pack import (CarmaModel, CarmaSample, Car1Sample, car1_process, car1_process_batch,  # noqa: F401
                         carma_process, carma_process_batch, carma_variance, get_ar_roots, power_spectrum)

__all__ = ["Context", "kfilter_carma", "kfilter_carma_batch", "kfilter_car1", "CarmaError", "CarmaDeviceError", "CarmaModel",
           "CarmaSample", "Car1Sample", "get_ar_roots", "power_spectrum", "carma_variance", "car1_process",
           "carma_process", "carma_process_batch", "car1_process_batch"]
