"""Prototype (numpy float64, round 6) of the TWO-SIDED filter: the serial chain of kfilter.cpp:189-215 cut in two.

The state z of the real modal recursion (real_modal.py) is a stationary Gauss-Markov process with covariance V, so it
is Markov in reversed time as well, with transition V F^T V^-1.  In the dual coordinates u = V^-1 z that is

    u_k = F_k^T u_{k+1} + noise ,  Cov u = V^-1 ,  y_k = h.z_k = (V h).u_k = c.u_k

-- the SAME recursion with h and c = V h exchanged (c' = V^-1 c = h, s0' = c.V^-1 c = h.c = s0), the rotation sense
reversed (F^T: omega -> conj omega) and the data taken last to first.  V never appears.

Forward over data 0 .. m-1 gives  z_m | y_a ~ N(a, V + Da);  backward over data n-1 .. m gives  u_m | y_b ~ N(beta,
V^-1 + Db), both propagated to ONE meeting time.  With alpha, beta as random variables (functions of the data):
Cov alpha = -Da, Cov beta = -Db, E[alpha beta^T] = Da E[u z^T] Db = Da Db (the halves are independent given z_m), and
alpha / beta are sufficient for z_m, so with N = I - Da Db

    log p(y) = l_a + l_b - 1/2 log det N + beta.N^-1 a + 1/2 beta.N^-1 Da beta + 1/2 (Db a).N^-1 a            (merge)

-- one p x p factorisation per evaluation, again without V.  (N has its eigenvalues in (0, 1]: -Da < V, -Db < V^-1.)

What the DEVICE evaluates is the same quantity through a congruence (merge_chol): with X = -Da = L L^T, Y = -Db,
W = I - L^T Y L = C C^T (symmetric, eigenvalues in (0, 1])

    log p(y) = l_a + l_b - 1/2 log det W + beta.a - 1/2 a.Y a - 1/2 |C^-1 L^T (Y a - beta)|^2

because in modal coordinates of nearly coincident roots X has entries ~ 1 / separation^2 which cancel in the product X Y,
and the plain N loses what the two recursions kept (roots 1e-6 apart: 1.5e-3 from the exact value, the one-pass filter
3e-6, merge_chol 4e-6).  X = L L^T takes DIAGONAL PIVOTING: X is numerically rank deficient as a rule (a half of the series
says nothing about modes that have decayed by the meeting time; the two coordinates of a pair can carry one direction), and
in a fixed order a pivot at rounding level met before an informative one grows into it.  lane_merge_chol is the computation
as the lanes do it (carma_pipew.h pipew_merge): X read symmetric bit for bit and equilibrated by powers of two, THRESHOLD
pivoting (coordinates in their own order, but only those above the level's threshold 4e-3, 4e-6, ...: within 1e3 of diagonal
pivoting's guarantee, and no index is a run-time value), right-looking l_i l_j updates with both factors from the pivot column
(so that the Schur complement stays symmetric), what is left at rounding level dropped, the border row v = L^T (Y a - beta)
riding along in the factorisation of W.

loglik_two_sided(...) below is the plain recursion on both sides (loglik_std of lazy_frame.py); the window / lane forms of
the device run the same two recursions chunk by chunk."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from real_modal import real_model, phi  # noqa: E402


def half_filter(om, pairs, p, h, c, s0, tt, yc, e, t_meet, transpose):
    """Recursion over the data (tt increasing or decreasing), then the prediction to t_meet.  Returns (l, D, z)."""
    D = np.zeros((p, p))
    z = np.zeros(p)
    ll = 0.0

    def F(dt):
        f = phi(om, pairs, abs(dt), p)
        return f.T if transpose else f

    for k in range(tt.size):
        if k > 0:
            f = F(tt[k] - tt[k - 1])
            D = f @ D @ f.T
            z = f @ z
        w = D @ h
        var = s0 + e[k] + h @ w
        kk = w + c
        innov = yc[k] - h @ z
        ll += -0.5 * np.log(var) - 0.5 * innov * innov / var
        D = D - np.outer(kk, kk) / var
        z = z + kk * (innov / var)
    if tt.size:
        f = F(t_meet - tt[-1])
        D = f @ D @ f.T
        z = f @ z
    return ll, D, z


def merge(Da, a, Db, beta):
    """log of  integral N(z; a, V + Da) N(V^-1 z; beta, V^-1 + Db) |V^-1| / N(z; 0, V) dz  (see the module text)."""
    p = a.size
    N = np.eye(p) - Da @ Db
    sign, logdet = np.linalg.slogdet(N)
    x1 = np.linalg.solve(N, a)
    x2 = np.linalg.solve(N, Da @ beta)
    return -0.5 * logdet + beta @ x1 + 0.5 * beta @ x2 + 0.5 * (Db @ a) @ x1


def chol_psd(X, tau=1e-15):
    """X = L L^T (columns in pivot order) with diagonal pivoting: the largest remaining diagonal is the pivot; what is left when
    the pivots reach rounding level (<= tau x the largest original diagonal) is dropped."""
    p = X.shape[0]
    A = X.copy()
    L = np.zeros((p, p))
    done = np.zeros(p, bool)
    top = max(np.max(np.diag(X)), 0.0)
    for m in range(p):
        dgs = np.where(done, -np.inf, np.diag(A))
        k = int(np.argmax(dgs))
        if not dgs[k] > tau * top:
            break
        L[:, m] = np.where(done, 0.0, A[:, k] / np.sqrt(dgs[k]))
        A -= np.outer(L[:, m], L[:, m])
        done[k] = True
    return L


def merge_chol(Da, a, Db, beta):
    """The same quantity as merge() through the congruence by the Cholesky factor of X = -Da (see the module text)."""
    X, Y = -0.5 * (Da + Da.T), -Db
    d = 1.0 / np.sqrt(np.where(np.diag(X) > 0, np.diag(X), 1.0))
    L = chol_psd(X * d[:, None] * d[None, :]) / d[:, None]      # (equilibrated, so that the threshold has a meaning)
    T = Y @ L
    W = np.eye(a.size) - L.T @ T
    v = T.T @ a - L.T @ beta
    C = np.linalg.cholesky(W)
    s = np.linalg.solve(C, v)
    return -np.sum(np.log(np.diag(C))) + beta @ a - 0.5 * a @ (Y @ a) - 0.5 * s @ s


LEVELS = (4e-3, 4e-6, 4e-9, 4e-12, 4e-15)


def lane_merge_chol(Da, a, Db, beta, stats=None):
    """merge_chol in pipew_merge's lane layout: lane j holds column j of the Schur complement and, once it has been taken, column j
    of L.  THIS function pivots by THRESHOLD LEVELS: the coordinates are taken in their own order, but only those whose remaining
    diagonal is above the level's threshold; the others wait for a later level.  A pivot is then within 1e3 of everything taken
    after it (the guarantee of diagonal pivoting up to that factor) and every broadcast comes from a lane known in advance.  The
    device tried this form (robust, but 3.6 k cycles for the loop against 2.6 k: profiles/r06/merge_variants_v1.txt) and kept plain
    DIAGONAL pivoting -- the largest remaining diagonal by a DPP maximum, its column by ds_bpermute, same equilibration, same rank
    threshold (4e-15) -- which is chol_psd / merge_chol above; this emulation stays as the record of the variant and as a second
    route to the same numbers in tests/test_two_sided_proto.py."""
    P = a.size
    A = np.array([[-(Da[i, j] if i >= j else Da[j, i]) for i in range(P)] for j in range(P)])      # A[j][i] = X_ij, lane j
    kf = np.array([[Db[i, j] for i in range(P)] for j in range(P)])
    # equilibration by powers of two (X <- D X D, Y <- D^-1 Y D^-1, a <- D a, beta <- D^-1 beta: every term is invariant), so that
    # the levels have a meaning: the diagonal of X is in [1, 4) wherever it is positive
    sj = np.array([-(int(np.frexp(A[j][j])[1]) >> 1) if A[j][j] > 0 and np.isfinite(A[j][j]) else 0 for j in range(P)])
    A = np.array([[np.ldexp(A[j][i], int(sj[i] + sj[j])) for i in range(P)] for j in range(P)])
    kf = np.array([[np.ldexp(kf[j][i], int(-sj[i] - sj[j])) for i in range(P)] for j in range(P)])
    a = np.ldexp(a, sj)
    beta = np.ldexp(beta, -sj)
    done = np.zeros(P, bool)
    rs = np.zeros(P)
    alive = np.zeros((P, P), bool)
    order = []
    for thr in LEVELS:
        for k in range(P):
            dk = A[k][k]
            if done[k] or not dk > thr:
                continue
            rs[k] = 1.0 / np.sqrt(dk)
            alive[k] = ~done                                  # rows alive at this pivot (k itself among them)
            lj = np.array([A[j][k] * rs[k] if (not done[j] and j != k) else 0.0 for j in range(P)])
            li = A[k] * rs[k]
            for j in range(P):
                A[j] = A[j] - li * lj[j]
            done[k] = True
            order.append((k, dk))
    if stats is not None:
        stats.append(order)
    Lc = np.array([[A[j][i] * rs[j] if alive[j][i] else 0.0 for i in range(P)] for j in range(P)])      # Lc[j][i] = L_ij
    T = np.zeros((P, P))                                                                            # T[j][i] = (Db L)_ij
    for k in range(P):
        for j in range(P):
            T[j] += kf[k] * Lc[j][k]
    W = np.zeros((P, P + 1))
    for i in range(P):
        for j in range(P):
            W[j][i] = (1.0 if i == j else 0.0) + sum(Lc[i][k] * T[j][k] for k in range(P))
    dba = np.array([kf[j] @ a for j in range(P)])
    u = -beta - dba
    for j in range(P):
        W[j][P] = sum(Lc[j][k] * u[k] for k in range(P))
    piv, s2 = np.ones(P), np.zeros(P)
    for k in range(P):
        dk = W[k][k]
        piv[k], s2[k] = dk, W[k][P] ** 2 / dk
        f = np.array([W[j][k] / dk if j > k else 0.0 for j in range(P)])
        bc = W[k].copy()
        for i in range(k + 1, P + 1):
            for j in range(P):
                W[j][i] -= bc[i] * f[j]
    return -0.5 * np.sum(np.log(piv)) - 0.5 * np.sum(s2 - 2.0 * a * (0.5 * dba + beta))


def loglik_two_sided(t, y, yerr, theta, p, q, m=None, where="left", merge_fn=None):
    """m data on the forward side (default (n + 1) // 2, the device's split); meeting time t[m-1] ("left", the device's), t[m]
    ("right") or halfway ("mid")."""
    if merge_fn is None:
        merge_fn = merge_chol
    om, h, Vz, pairs = real_model(theta, p, q)
    n = t.size
    if m is None:
        m = (n + 1) // 2
    yc = y - theta[2]
    e = theta[1] * yerr ** 2
    c = Vz @ h
    s0 = h @ Vz @ h
    if m <= 0 or m >= n:
        return half_filter(om, pairs, p, h, c, s0, t, yc, e, t[-1], False)[0]
    tm = {"left": t[m - 1], "right": t[m], "mid": 0.5 * (t[m - 1] + t[m])}[where]
    la, Da, a = half_filter(om, pairs, p, h, c, s0, t[:m], yc[:m], e[:m], tm, False)
    lb, Db, beta = half_filter(om, pairs, p, c, h, s0, t[m:][::-1], yc[m:][::-1], e[m:][::-1], tm, True)
    return la + lb + merge_fn(Da, a, Db, beta)


def loglik_one_pass(t, y, yerr, theta, p, q):
    om, h, Vz, pairs = real_model(theta, p, q)
    return half_filter(om, pairs, p, h, Vz @ h, h @ Vz @ h, t, y - theta[2], theta[1] * yerr ** 2, t[-1], False)[0]


if __name__ == "__main__":
    import oracle as orc
    from carma_pack_amd.synth import theta_batch

    g = np.load(os.path.join(ROOT, "tests", "golden", "carma53_readme.npz"))
    t, y, yerr = g["t"], g["y"], g["yerr"]
    p, q = 5, 3
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    th = theta_batch(np.random.default_rng(2), N, p, q, t, y, theta_center=g["theta"][0])
    mdl = orc.OracleModel(t, y, yerr, p, q, max_stdev=10 * y.std())
    ref = mdl.logdensity_batch(th, ignore_prior=True)
    lp = np.array([mdl.log_prior(x) for x in th])
    for where in ("left", "mid", "right"):
        e1, e2 = [], []
        for i in range(N):
            if not np.isfinite(ref[i]):
                continue
            want = ref[i] - lp[i]
            with np.errstate(all="ignore"):
                a = loglik_one_pass(t, y, yerr, th[i], p, q)
                b = loglik_two_sided(t, y, yerr, th[i], p, q, where=where)
            e1.append(abs(a - want) / max(1.0, abs(want)))
            e2.append(abs(b - want) / max(1.0, abs(want)))
        e1, e2 = np.array(e1), np.array(e2)
        for nm, v in (("one pass", e1), ("two-sided " + where, e2)):
            print("%-18s n=%d median %.2e  99%% %.2e  max %.2e  >1e-10: %d" % (nm, v.size, np.median(v), np.quantile(v, 0.99), v.max(), np.sum(v > 1e-10)))
