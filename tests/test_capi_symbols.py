"""CPU-only: the C-ABI library loads and exports every symbol include/carma_mi355.h declares,
argument errors are reported, and compute calls fail loudly (no CPU fallback) without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "carma_mi355.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(carma_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import carma_pack_amd._lib as L
    syms = _declared_symbols()
    assert len(syms) >= 15
    dll = ctypes.CDLL(L.LIB_PATH)
    for s in syms:
        assert hasattr(dll, s), "libcarma_mi355.so does not export %s" % s
    assert sorted(L.EXPORTS) == syms, "carma_pack_amd/_lib.py EXPORTS out of sync with the header"


def test_no_cpu_fallback_without_gpu():
    import carma_pack_amd as cpa
    if cpa._lib.lib.carma_device_count() > 0:
        pytest.skip("a GPU is visible")
    t = np.arange(10.0)
    with pytest.raises(cpa.CarmaDeviceError):
        cpa.Context(t, np.sin(t), np.ones(10), 3, 1)
    with pytest.raises(cpa.CarmaError):
        cpa.kfilter_car1(t, np.sin(t), np.ones(10), 1.0, 0.1)
    # argument checks of the optimiser entry point come before any device work
    import ctypes as C
    x = np.zeros(4)
    assert cpa._lib.lib.carma_mle_batched(None, x.ctypes.data_as(C.c_void_p), 1, None, None, 10, 8, 1e-9, 1e-5, 1e-6, 1,
                                          x.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p), None, None, None) == -22


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under carma_pack_amd/ may reference it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "carma_pack_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.lower(), "%s mentions the oracle" % f
