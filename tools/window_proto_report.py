#!/usr/bin/env python3
"""Error table of the BLOCKED-window prototype (tests/tools/proto/blocked_window.py: a chunk of the co-rotating-frame recursion as
one symmetric elimination -- loglik_ldl -- and the same elimination one lane per datum with the columns of S as virtual lanes --
loglik_window) against the oracle: README fixture and perturbed parameter vectors, chunks cut by re-base data, the OGLE order
grid.  CPU only; the record is profiles/r05/blocked_proto_v1.txt (cited by tests/test_blocked_window_proto.py)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "tools", "proto"))
import oracle as orc
from blocked_window import loglik_ldl, loglik_window
from carma_pack_amd.synth import theta_batch


def errs(t, y, yerr, theta, p, q, **kw):
    m = orc.OracleModel(t, y, yerr, p, q, max_stdev=1e300)
    ref = m.logdensity(theta, ignore_prior=True)
    if not np.isfinite(ref):
        return None
    want = ref - m.log_prior(theta)
    out = []
    for f, mc in ((loglik_ldl, 16), (loglik_window, None)):
        st = []
        with np.errstate(all="ignore"):
            got = f(t, y, yerr, theta, p, q, mchunk=mc, stats=st, **kw)
        out.append((abs(got - want) / max(1.0, abs(want)), st[0]))
    return out


G = os.path.join(ROOT, "tests", "golden")
g = np.load(os.path.join(G, "carma53_readme.npz"))
t, y, e = g["t"], g["y"], g["yerr"]
print("case | vectors | worst error loglik_ldl | worst error loglik_window | re-base data (first vector)")
th = np.concatenate([g["theta"][:8], theta_batch(np.random.default_rng(3), 24, 5, 3, t, y, theta_center=g["theta"][0])])
r = [x for x in (errs(t, y, e, v, 5, 3) for v in th) if x]
print("README CARMA(5,3), fixture + perturbed | %d | %.1e | %.1e | %d" % (len(r), max(a[0][0] for a in r), max(a[1][0] for a in r), r[0][0][1]))
for lim in (0.0, 0.05, 0.5, 5.0):
    a = errs(t, y, e, g["theta"][0], 5, 3, lim_re=lim, lim_im=1e9)
    print("README, window limit |Re omega| x window = %g | 1 | %.1e | %.1e | %d" % (lim, a[0][0], a[1][0], a[0][1]))
og = np.load(os.path.join(G, "ogle_grid.npz"))
to, yo, eo = og["t"], og["y"], og["yerr"]
worst = [0.0, 0.0]; n = 0
for p in range(2, 8):
    for q in (0, p - 1):
        for x in og["p%dq%d_theta" % (p, q)][:2]:
            a = errs(to, yo - yo.mean(), eo, x, p, q)
            if a:
                n += 1; worst = [max(worst[0], a[0][0]), max(worst[1], a[1][0])]
print("OGLE-LMC-LPV-00007, p = 2 ... 7, q = 0 and p - 1, two vectors each | %d | %.1e | %.1e | -" % (n, worst[0], worst[1]))
