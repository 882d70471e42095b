"""One-off soak: every order (p, q < p), regular-cadence series and data in odd units, all launch shapes, GPU against the
oracle with the quad-precision arbiter (the -m gpu tests cover three orders of each; this covers all 27)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import carma_pack_amd as cpa
import oracle as orc
from carma_pack_amd.synth import prior_like_theta
from helpers import assert_parity, loglik_truth
bad = 0
for p in range(2, 8):
    for q in range(p):
        rng = np.random.default_rng(100 * p + q)
        n = 141
        t = 1.5 * np.arange(n, dtype=float); t[50:] += 20.0; t[100:] += 3.25
        for unit in (1.0, 1e-30, 1e30):
            y0 = 2.0 + np.sin(t / 5.0) + 0.3 * rng.standard_normal(n); e0 = np.full(n, 0.3)
            th = np.array([prior_like_theta(rng, p, q, t, y0) for _ in range(40)])
            th[:, 0] *= unit; th[:, 2] *= unit
            y, e = unit * y0, unit * e0
            ctx = cpa.Context(t, y, e, p, q)
            m = orc.OracleModel(t, y, e, p, q, max_stdev=ctx.prior()[0])
            want = m.logdensity_batch(th, ignore_prior=True)
            for B in (40, 3600, 72000):
                got = ctx.logdensity(np.tile(th, (B // 40, 1)), ignore_prior=True)[:40]
                try:
                    assert_parity(got, want, 1e-10, "p=%d q=%d unit=%g %s" % (p, q, unit, ctx.kernel_name(B)),
                                  arbiter=lambda i: loglik_truth(t, y, e, th[i], p, q)[0])
                except AssertionError as ex:
                    bad += 1
                    print("FAIL", ex, flush=True)
    print("p=%d done" % p, flush=True)
print("failures:", bad)
