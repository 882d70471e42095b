"""Fuzz of the log-density launcher: random orders, series lengths and batch sizes AROUND EVERY DISPATCH THRESHOLD (one row
/ workgroup / wave more or less than a launch shape takes), prior-like parameter vectors with the bounds on or off.
Checked per case: the copies of a parameter vector give the same bits wherever they sit in the launch (first, last,
partial workgroup), and the distinct vectors agree with the oracle (1e-10, or never further from the quad-precision
value).  Run on the GPU box:  python tools/fuzz_dispatch.py [cases] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import carma_pack_amd as cpa
import oracle as orc
from helpers import assert_parity, irregular_series, loglik_truth, prior_like_theta
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
EDGES = [1, 4, 512, 1024, 1536, 2048, 3072, 4096, 8192, 16384, 24576, 32768, 49152, 65536]
seen, fails = {}, 0
t00 = time.time()
for case in range(ncase):
    p = int(rng.integers(1, 8))
    q = int(rng.integers(0, p)) if p > 1 else 0
    n = int(rng.choice([2, 5, 7, 8, 9, 15, 16, 17, 31, 33, 50, 97, 130, 270]))
    if rng.random() < 0.12:                                           # round 6: the two-sided kernels' series limits (LDS / global memory)
        n = int(rng.choice([437, 1024, 1025, 2999, 5000, 5001, 6500]))
    B = int(rng.choice(EDGES)) + int(rng.integers(-3, 4))
    if rng.random() < 0.15:
        B = int(rng.integers(1, 70000))
    B = max(B, 1)
    if n > 1000:
        B = min(B, int(rng.choice([3, 511, 513, 1537, 2100])))
    ign = bool(rng.random() < 0.4) and p > 1
    t, y, yerr = irregular_series(n, seed=int(rng.integers(1, 10 ** 6)))
    if rng.random() < 0.25 and n > 4:                               # a regular stretch: repeated time steps
        t = np.cumsum(np.r_[t[0], np.where(rng.random(n - 1) < 0.8, 1.5, rng.uniform(0.5, 4.0, n - 1))])
    ctx = cpa.Context(t, y, yerr, p, q)
    m = orc.OracleModel(t, y, yerr, p, q, max_stdev=ctx.prior()[0])
    K = 37
    th = np.array([prior_like_theta(rng, p, q, t, y) for _ in range(K)])
    big = np.tile(th, (B // K + 1, 1))[:B]
    name = ctx.kernel_name(B)
    seen[name.split("<")[0]] = seen.get(name.split("<")[0], 0) + 1
    if os.environ.get("FUZZ_VERBOSE"):                                # (a memory fault takes the process down: the case is on record first)
        print("case %d: CARMA(%d,%d) n=%d B=%d ignore_prior=%s %s" % (case, p, q, n, B, ign, name), flush=True)
    try:
        got = ctx.logdensity(big, ignore_prior=ign) if p > 1 else ctx.logdensity(big)
        assert got.shape == (B,)
        assert np.array_equal(got, np.tile(got[:K], B // K + 1)[:B], equal_nan=True), "copies differ"
        kk = min(K, B)
        want = m.logdensity_batch(th[:kk], ignore_prior=ign) if p > 1 else m.logdensity_batch(th[:kk])
        roots = [np.asarray(orc.ar_roots(v, p)) for v in th[:kk]] if p > 1 else []
        dup = np.array([np.min(np.abs(r[:, None] - r[None, :]) + np.eye(p)) == 0.0 for r in roots]) if p > 1 else np.zeros(kk, bool)
        ok = ~dup
        assert_parity(got[:kk][ok], want[ok], 1e-10, "case", arbiter=lambda i: loglik_truth(t, y, yerr, th[np.flatnonzero(ok)[i]], p, q)[0],
                      max_arbitrated=kk)
    except AssertionError as ex:
        fails += 1
        print("FAILED case %d: CARMA(%d,%d) n=%d B=%d ignore_prior=%s %s: %s" % (case, p, q, n, B, ign, name, str(ex)[:300]), flush=True)
print("%d cases in %.0f s, %d failed; launches by kernel: %s" % (ncase, time.time() - t00, fails, seen))
