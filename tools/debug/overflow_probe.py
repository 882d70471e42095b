"""Where the reference's MA coefficients leave the double range (hot chains drift there: the MA parameters are unbounded and
the likelihood is flat in that direction): finite patterns of the device kernels and the oracle side by side."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import carma_pack_amd as cpa
import oracle as orc
from helpers import irregular_series, prior_like_theta
rng = np.random.default_rng(5)
for (p, q) in ((5, 3), (6, 5), (7, 6), (4, 2), (3, 2), (2, 1)):
    t, y, yerr = irregular_series(150, seed=p)
    ctx = cpa.Context(t, y, yerr, p, q)
    m = orc.OracleModel(t, y, yerr, p, q, max_stdev=ctx.prior()[0])
    B = 4096
    th = np.array([prior_like_theta(rng, p, q, t, y) for _ in range(B)])
    npair = max(q // 2, 1)
    for b in range(B):
        tot = rng.uniform(-460.0, -250.0)                # log of the product of the MA roots: the coefficients reach e^-tot
        w = rng.dirichlet(np.ones(npair))
        if q >= 2:
            for i in range(q // 2):
                th[b, 3 + p + 2 * i] = tot * w[i]
                th[b, 3 + p + 2 * i + 1] = rng.uniform(-3, 3) + 0.5 * tot * w[i] * rng.uniform(0.0, 1.2)
        else:
            th[b, 3 + p] = tot
    want = m.logdensity_batch(th, nthreads=os.cpu_count() or 8)
    for reps in (1, 12):
        got = ctx.logdensity(np.tile(th, (reps, 1)))[:B]
        fg, fw = np.isfinite(got), np.isfinite(want)
        lm = np.array([np.log10(np.max(np.abs(orc.ma_coefs(x, p, q)))) for x in th])
        both = fg & fw
        rel = np.abs(got[both] - want[both]) / np.abs(want[both])
        print("CARMA(%d,%d) %s: oracle finite %d, device finite %d, device only %d (log10 max|ma| %s), oracle only %d (%s); both finite: %d, rel diff > 1e-8: %d, log10 max|ma| of those from %.0f" % (
            p, q, ctx.kernel_name(reps * B), fw.sum(), fg.sum(), (fg & ~fw).sum(), np.round(np.sort(lm[fg & ~fw])[[0, -1]], 1) if (fg & ~fw).any() else "-",
            (fw & ~fg).sum(), np.round(np.sort(lm[fw & ~fg])[[0, -1]], 1) if (fw & ~fg).any() else "-", both.sum(), (rel > 1e-8).sum(),
            lm[both][rel > 1e-8].min() if (rel > 1e-8).any() else np.nan), flush=True)
        if (rel > 1e-8).any():
            w = np.flatnonzero(both)[rel > 1e-8]
            tr = np.array([orc.truth_logdensity(t, y, yerr, th[i], p, q)[0] for i in w])
            print("      of those: device nearer the exact value in %d, worst device rel err %.1e, worst oracle rel err %.1e" % (
                (np.abs(got[w] - tr) <= np.abs(want[w] - tr)).sum(), np.max(np.abs(got[w] - tr) / np.abs(tr)), np.max(np.abs(want[w] - tr) / np.abs(tr))))
        print("      finite oracle: log10 max|ma| up to %.1f; non-finite oracle from %.1f" % (lm[fw].max(), lm[~fw].min() if (~fw).any() else np.nan))
