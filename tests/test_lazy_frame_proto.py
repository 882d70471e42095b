"""CPU check of the algebra behind the latency-regime kernel (carma_pipe3l.h): the covariance recursion in a frame that
co-rotates with the transition (numpy prototype tests/tools/proto/lazy_frame.py) gives the oracle's log-likelihood,
whatever the length of the windows between re-bases -- one datum (every step re-based = the stepwise recursion) up to
the whole series."""
import os
import sys

import numpy as np
import pytest

import oracle as orc

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "proto"))
from lazy_frame import loglik_lazy, loglik_std  # noqa: E402
from carma_pack_amd.synth import theta_batch  # noqa: E402


@pytest.fixture(scope="module")
def readme(golden_dir):
    return np.load(os.path.join(golden_dir, "carma53_readme.npz"))


def test_corotating_recursion_equals_the_oracle(readme):
    g = readme
    t, y, yerr = g["t"], g["y"], g["yerr"]
    p, q = 5, 3
    th = theta_batch(np.random.default_rng(3), 10, p, q, t, y, theta_center=g["theta"][0])
    m = orc.OracleModel(t, y, yerr, p, q, max_stdev=10 * y.std())
    ref = m.logdensity_batch(th, ignore_prior=True)
    nchecked = 0
    for i in range(th.shape[0]):
        if not np.isfinite(ref[i]):
            continue
        want = ref[i] - m.log_prior(th[i])
        std = loglik_std(t, y, yerr, th[i], p, q)
        nreb = {}
        for name, kw in (("stepwise", dict(lim_re=0.0, lim_im=0.0, maxwin=0)),             # every datum re-based
                         ("kernel", dict(lim_re=200.0, lim_im=256.0, maxwin=10 ** 9)),      # the kernel's window limits
                         ("one window", dict(lim_re=1e300, lim_im=1e300, maxwin=10 ** 9))):  # never re-based
            st = []
            with np.errstate(all="ignore"):
                got = loglik_lazy(t, y, yerr, th[i], p, q, stats=st, **kw)
            nreb[name] = st[0]
            if name == "one window" and not np.isfinite(got):
                continue                        # scale factors e^{|Re omega| 900} overflow: what the re-base is for
            assert abs(got - want) <= 1e-10 * max(1.0, abs(want)), (i, name, got, want)
            assert abs(got - std) <= 2e-11 * max(1.0, abs(std)), (i, name)
        assert nreb["stepwise"] == t.size - 1 and nreb["one window"] == 0 and nreb["kernel"] < 60
        nchecked += 1
    assert nchecked >= 8
