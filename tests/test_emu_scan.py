"""CPU-only: the time-parallel (associative scan) form of the filter, carma_pack_amd/csrc/carma_scan.h,
executed lane by lane on the host (tests/emu) against the oracle.  The model in real modal coordinates is
built here with numpy (the device builds it in its set-up code)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import emu_build as emu
import oracle as orc
from helpers import irregular_series, prior_like_theta

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "proto"))
from scan_real import real_model  # noqa: E402


def scan_loglik(t, y, yerr, theta, p, q, nlanes=64):
    om, h, Vz, pairs = real_model(theta, p, q)
    cpx = np.zeros(p, dtype=np.int32)
    for r, c in pairs:
        if c:
            cpx[r] = cpx[r + 1] = 1
    lib = emu.lib()
    lib.emu_scan_loglik.restype = C.c_double
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    s = emu.pack_series(t, y, yerr)
    wre, wim = np.ascontiguousarray(om.real), np.ascontiguousarray(om.imag)
    h, Vz = np.ascontiguousarray(h), np.ascontiguousarray(Vz)
    return lib.emu_scan_loglik(p, wre.ctypes.data_as(dp), wim.ctypes.data_as(dp), cpx.ctypes.data_as(ip),
                               h.ctypes.data_as(dp), Vz.ctypes.data_as(dp), C.c_double(theta[2]), C.c_double(theta[1]),
                               s.ctypes.data_as(dp), t.size, nlanes)


@pytest.mark.parametrize("p,q,n,nlanes", [(5, 3, 270, 64), (5, 3, 270, 40), (2, 1, 100, 64), (3, 2, 77, 16), (4, 0, 300, 64),
                                          (6, 5, 130, 64), (5, 0, 64, 64), (5, 4, 65, 64)])
def test_scan_vs_oracle(p, q, n, nlanes, golden_dir):
    if n == 270:
        g = np.load(os.path.join(golden_dir, "carma53_readme.npz"))
        t, y, yerr, th = g["t"], g["y"], g["yerr"], g["theta"][:8]
    else:
        t, y, yerr = irregular_series(n, seed=n + p)
        rng = np.random.default_rng(10 * p + q)
        th = np.array([prior_like_theta(rng, p, q, t, y) for _ in range(6)])
    m = orc.OracleModel(t, y, yerr, p, q)
    want = m.logdensity_batch(th, ignore_prior=True)
    for i, theta in enumerate(th):
        if not np.isfinite(want[i]):
            continue
        got = scan_loglik(t, y, yerr, theta, p, q, nlanes) + m.log_prior(theta)
        assert abs(got - want[i]) <= 1e-10 * abs(want[i]), (p, q, n, i, got, want[i])
