"""Sanity check at n = 10^5 (beyond every BASELINE config): the latency-regime pipeline against the oracle.
Run on the GPU box:  python tests/tools/long_series_check.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import carma_pack_amd as cpa
import oracle as orc
from carma_pack_amd.synth import irregular_series, prior_like_theta

for p, q, n in ((5, 3, 100000), (7, 6, 100000), (2, 0, 100000)):
    t, y, yerr = irregular_series(n, seed=9)
    rng = np.random.default_rng(90 + p)
    th = np.array([prior_like_theta(rng, p, q, t, y) for _ in range(16)])
    ctx = cpa.Context(t, y, yerr, p, q)
    m = orc.OracleModel(t, y, yerr, p, q, max_stdev=ctx.prior()[0])
    t0 = time.perf_counter(); got = ctx.logdensity(th, ignore_prior=True); tg = time.perf_counter() - t0
    t0 = time.perf_counter(); want = m.logdensity_batch(th, ignore_prior=True, nthreads=16); tc = time.perf_counter() - t0
    fin = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), fin)
    rel = np.abs(got[fin] - want[fin]) / np.abs(want[fin])
    plain = ctx.logdensity(np.tile(th, (600, 1)), ignore_prior=True)[:16]       # throughput kernel: stepwise rotation
    relp = np.abs(plain[fin] - want[fin]) / np.abs(want[fin])
    print("p=%d q=%d n=%d: %d finite, max rel diff %.2e, median %.2e (stepwise kernel: max %.2e median %.2e); gpu %.1f ms, "
          "oracle (16 threads) %.2f s" % (p, q, n, fin.sum(), rel.max(), np.median(rel), relp.max(), np.median(relp), tg * 1e3, tc))
