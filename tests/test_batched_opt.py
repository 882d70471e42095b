"""CPU-only: the numpy PROTOTYPE (tests/tools/batched_opt_proto.py) of the lock-step batched L-BFGS that the library
runs natively (carma_mle.hip); test_gpu_api.py holds the native optimiser to it start by start."""
import numpy as np
from scipy.optimize import minimize

import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
from batched_opt_proto import minimize_batched  # noqa: E402


def _rosen(x):
    x = np.atleast_2d(x)
    return np.sum(100.0 * (x[:, 1:] - x[:, :-1] ** 2) ** 2 + (1.0 - x[:, :-1]) ** 2, axis=1)


def test_batched_rosenbrock_matches_scipy():
    rng = np.random.default_rng(0)
    x0 = rng.uniform(-1.5, 1.5, (24, 4))
    bounds = [(-2.0, 2.0)] * 4
    res = minimize_batched(_rosen, x0, bounds, maxiter=500)
    for r, s in zip(res, x0):
        ref = minimize(lambda v: _rosen(v)[0], s, method="L-BFGS-B", bounds=bounds)
        assert r.fun <= ref.fun + 1e-5 or r.fun < 1e-5 or abs(r.fun - ref.fun) < 1e-3
    assert sum(r.fun < 1e-6 for r in res) >= 18


def test_bounds_and_infeasible_regions():
    # minimum outside the box -> solution on the bound; NaN region handled as infeasible
    def f(x):
        x = np.atleast_2d(x)
        v = np.sum((x - 3.0) ** 2, axis=1)
        v[x[:, 0] < -0.5] = np.nan
        return v
    res = minimize_batched(f, np.zeros((5, 3)), [(-1.0, 1.0), (None, None), (0.0, 2.5)])
    for r in res:
        np.testing.assert_allclose(r.x, [1.0, 3.0, 2.5], atol=1e-4)
        assert r.success
