#!/usr/bin/env python3
"""Why the merge of the two-sided filter factors X = -Da with DIAGONAL PIVOTING (carma_pipew.h, pipew_merge): on prior-like
parameter vectors of the README CARMA(5,3) case, the pivots of X = L L^T taken in the coordinates' own (root) order and with the
largest remaining diagonal first, and what each does to the log-likelihood (numpy prototype of the device's arithmetic,
tests/tools/proto/two_sided.py; errors against the oracle's one-pass value).  CPU only."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "tools", "proto"))
import two_sided as ts
import oracle as orc
from carma_pack_amd.synth import theta_batch

g = np.load(os.path.join(ROOT, "tests", "golden", "carma53_readme.npz"))
t, y, e = g["t"], g["y"], g["yerr"]
p, q = 5, 3
TAU = 4e-15
piv_log = {}


def chol_order(X, order, tag):
    """X = L L^T in a given order of the coordinates (None: diagonal pivoting), pivots below TAU x the largest diagonal dropped"""
    n = X.shape[0]
    A = X.copy(); L = np.zeros((n, n)); done = np.zeros(n, bool); pv = []
    for m in range(n):
        if order is None:
            dgs = np.where(done, -np.inf, np.diag(A)); k = int(np.argmax(dgs))
        else:
            k = order[m]; dgs = np.diag(A)
        pv.append(float(dgs[k]))
        if not dgs[k] > TAU:
            if order is None:
                break
            done[k] = True
            continue
        L[:, m] = np.where(done, 0.0, A[:, k] / np.sqrt(dgs[k]))
        A -= np.outer(L[:, m], L[:, m])
        done[k] = True
    piv_log[tag] = pv
    return L


def merge_with(order, tag):
    def f(Da, a, Db, beta):
        X, Y = -0.5 * (Da + Da.T), -Db
        d = 1.0 / np.sqrt(np.where(np.diag(X) > 0, np.diag(X), 1.0))
        L = chol_order(X * d[:, None] * d[None, :], order, tag) / d[:, None]
        T = Y @ L
        W = np.eye(a.size) - L.T @ T
        v = T.T @ a - L.T @ beta
        C = np.linalg.cholesky(W)
        s = np.linalg.solve(C, v)
        return -np.sum(np.log(np.diag(C))) + beta @ a - 0.5 * a @ (Y @ a) - 0.5 * s @ s
    return f


th = theta_batch(np.random.default_rng(7), 600, p, q, t, y, theta_center=g["theta"][0], frac_post=0.0)
m = orc.OracleModel(t, y, e, p, q, max_stdev=1e300)
rows = []
for k, x in enumerate(th):
    ref = m.logdensity(x, ignore_prior=True)
    if not np.isfinite(ref):
        continue
    want = ref - m.log_prior(x)
    out = {}
    with np.errstate(all="ignore"):
        for tag, order in (("root order", list(range(p))), ("reversed", list(range(p - 1, -1, -1))), ("pivoted", None)):
            try:
                got = ts.loglik_two_sided(t, y, e, x, p, q, merge_fn=merge_with(order, tag))
            except np.linalg.LinAlgError:
                got = np.nan
            out[tag] = (abs(got - want) / max(1.0, abs(want)), list(piv_log.get(tag, [])))
    rows.append((k, out))
print("X = -Da factored after equilibration (unit diagonal where positive); pivots below %.0e dropped; %d prior-like vectors" % (TAU, len(rows)))
for tag in ("root order", "reversed", "pivoted"):
    errs = np.array([r[1][tag][0] for r in rows])
    print("%-11s worst error %.1e, entries beyond 1e-10: %d, not finite: %d" % (tag, np.nanmax(errs), int(np.nansum(errs > 1e-10)), int(np.sum(~np.isfinite(errs)))))
worst = sorted(rows, key=lambda r: -np.nan_to_num(r[1]["root order"][0], nan=1.0))[:6]
for k, out in worst:
    print("vector %d:" % k)
    for tag in ("root order", "reversed", "pivoted"):
        print("   %-11s error %.1e   pivots %s" % (tag, out[tag][0], " ".join("%.1e" % v for v in out[tag][1])))
