"""Shared helpers for the test-suite (synthetic series + theta generators)."""
import numpy as np


from carma_pack_amd.synth import irregular_series, log_quads_from_roots, prior_like_theta, theta_batch  # noqa: F401,E402

# ---- the parity allowances a run CONSUMED ---------------------------------------------------------
# Every clause below that lets an entry pass on something other than "within 1e-10 of the oracle" records its use here:
# which clause, in which test, how many entries used it against how many were allowed, and the worst device / oracle
# distance from the quad-precision value it saw.  tests/conftest.py writes the list to gpurun_out/parity_allowances.json at
# the end of the session; tools/allowance_table.py prints the table DESIGN.md section 4 carries.
ALLOWANCES = []


def record_allowance(clause, what, used, allowed, population=None, worst_device=None, worst_oracle=None, cond=None):
    import os
    ALLOWANCES.append(dict(clause=clause, test=os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0], what=str(what),
                           used=int(used), allowed=int(allowed), population=None if population is None else int(population),
                           worst_device=None if worst_device is None else float(worst_device),
                           worst_oracle=None if worst_oracle is None else float(worst_oracle),
                           cond=None if cond is None else float(cond)))


def loglik_truth(t, y, yerr, theta, p, q):
    """(log-likelihood + log prior, log-likelihood) of the reference's formulas to > 20 digits: the arbiter of the parity
    tests.  Quad-precision restatement in C (oracle/carma_truth_q.c), itself pinned against the 50-digit mpmath version
    (tests/mp_truth.py) in tests/test_oracle_golden.py; ~100x faster, so EVERY entry that needs it can be arbitrated."""
    import oracle as orc
    return orc.truth_logdensity(t, y, yerr, theta, p, q)


def oracle_noise_scale(model, t, y, yerr, theta, p, q, nulp=3):
    """The reference's OWN error scale at theta: the largest distance of the oracle from the exact (quad-precision) value over
    theta and its +-1..nulp-ulp neighbours in the AR parameters.  Where roots nearly coincide (a quadratic factor with two real
    roots 6e-4 apart, cond(EigenMat) 1e6: tools/soak_pt_lane.py found one) every double-precision implementation's error jumps
    between 1e-8 and 3e-7 from one ulp of theta to the next -- which of two implementations is nearer at ONE theta is luck, the
    scale is not."""
    theta = np.asarray(theta, dtype=float)
    worst = 0.0
    for j in range(3, 3 + (p if p > 1 else 1)):
        for sgn in (-1.0, 1.0):
            x = theta.copy()
            for _ in range(nulp):
                x[j] = np.nextafter(x[j], sgn * np.inf)
                truth = loglik_truth(t, y, yerr, x, p, q)[0]
                v = model.logdensity(x)
                if np.isfinite(v) and np.isfinite(truth):
                    worst = max(worst, abs(v - truth) / abs(truth))
    return worst


def assert_parity(got, want, rtol=1e-10, what="", arbiter=None, max_arbitrated=None, arb_factor=1.0, max_arb_frac=0.01, noise_scale=None):
    """north_star bar: |got-want| <= 1e-10 |want| where finite; identical -inf/NaN pattern.

    Where roots cluster (cond(EigenMat) >~ 1e6; the prior admits roots 1e-4 apart) the REFERENCE's arithmetic --
    an LU solve of the Vandermonde system and p-term sums that cancel -- is itself 1e-10 ... 1e-3 away from the exact
    value of its own formulas, so its restatement cannot be the yardstick there.  If `arbiter(i)` is given it returns
    the exact value of entry i (helpers.loglik_truth: quad precision, > 20 digits); an entry that differs from the
    oracle by more than rtol passes only when the GPU is within rtol of the exact value or NO FURTHER from it than the
    oracle is (arb_factor 1.0: "never worse than the reference").

    Arbitration is the exception, not a second bar: at most `max_arb_frac` of the finite entries (1 %; never fewer than
    3 entries) -- or `max_arbitrated` entries when a test states its own number -- may need it, and the count is
    printed.  A regression that sent half a batch to the arbiter fails here."""
    got, want = np.asarray(got, dtype=float), np.asarray(want, dtype=float)
    assert got.shape == want.shape
    fin = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), fin), "%s: finite pattern differs at %s" % (
        what, np.flatnonzero(np.isfinite(got) != fin)[:10])
    assert np.array_equal(np.isneginf(got), np.isneginf(want)), "%s: -inf pattern differs" % what
    if not fin.any():
        return 0.0
    idx = np.flatnonzero(fin)
    rel = np.abs(got[fin] - want[fin]) / np.abs(want[fin])
    bad = idx[rel > rtol]
    if bad.size and arbiter is not None:
        limit = max_arbitrated if max_arbitrated is not None else max(3, int(np.ceil(max_arb_frac * idx.size)))
        assert bad.size <= limit, (
            "%s: %d of %d finite entries differ from the oracle by more than %.0e and would need arbitration "
            "(allowed: %d); worst %.3e" % (what, bad.size, idx.size, rtol, limit, rel.max()))
        print("%s: %d of %d finite entries go to the arbiter (allowed %d)" % (what, bad.size, idx.size, limit))
        wd, wo, nnoise = 0.0, 0.0, 0
        for i in bad:
            truth = arbiter(int(i))
            eg, eo = abs(got[i] - truth), abs(want[i] - truth)
            if eg > max(rtol * abs(truth), arb_factor * eo) and noise_scale is not None:
                # further than the oracle AT THIS theta: is it within the oracle's own error scale around it?
                ns = noise_scale(int(i))
                print("%s: entry %d: gpu err %.2e, oracle err %.2e here, up to %.2e within 3 ulp of theta" % (
                    what, i, eg / abs(truth), eo / abs(truth), ns))
                # (twice the largest of 6 p neighbours: a handful of samples underestimates a scale)
                eo = max(eo, 2.0 * ns * abs(truth) / max(arb_factor, 1e-300))
                nnoise += 1
            wd, wo = max(wd, eg / abs(truth)), max(wo, abs(want[i] - truth) / abs(truth))
            assert eg <= max(rtol * abs(truth), arb_factor * eo), (
                "%s: entry %d differs from the oracle by %.2e and is further from the exact value "
                "(gpu err %.2e, oracle err %.2e)" % (what, i, abs(got[i] - want[i]) / abs(want[i]),
                                                     eg / abs(truth), eo / abs(truth)))
            print("%s: entry %d arbitrated: gpu err %.2e, oracle err %.2e vs the exact (quad-precision) value" % (
                what, i, eg / abs(truth), eo / abs(truth)))
        record_allowance("arbiter (device no further from the exact value than %.2f x the oracle)" % arb_factor, what, bad.size,
                         limit, idx.size, wd, wo)
        if nnoise:
            record_allowance("oracle noise scale (+-3 ulp of theta, doubled)", what, nnoise, limit, idx.size, wd, wo)
        ok = np.ones(rel.size, dtype=bool)
        ok[np.isin(idx, bad)] = False
        return float(rel[ok].max()) if ok.any() else 0.0
    assert rel.max() <= rtol, "%s: max rel err %.3e at %d (got %r want %r)" % (
        what, rel.max(), idx[rel.argmax()], got[fin][rel.argmax()], want[fin][rel.argmax()])
    return float(rel.max())


def cond_eigenmat(theta, p):
    """2-norm condition number of the reference's EigenMat (the Vandermonde matrix of the AR roots, kfilter.cpp:144-149) --
    what its LU solve (:157-158) amplifies rounding by."""
    import oracle as orc
    w = np.asarray(orc.ar_roots(np.asarray(theta, dtype=float), p))
    E = np.vander(w, p, increasing=True).T
    with np.errstate(all="ignore"):
        try:
            return float(np.linalg.cond(E))
        except np.linalg.LinAlgError:
            return np.inf


def parity_census(got, want, thetas, p, labels, bounds, arbiter, rtol=1e-10, what="", arb_factor=1.15):
    """Parity BY CLASS instead of one allowance for everything: `labels[i]` names the class of entry i (e.g. "cold" / "hot"),
    `bounds[class]` is the largest fraction of that class's finite entries that may differ from the oracle by more than rtol.
    Every such entry is arbitrated against the quad-precision value (the device must be within rtol of it, or no further from
    it than arb_factor x the oracle's own distance: sampler states with cond(EigenMat) ~ 1e5 exist on which BOTH
    double-precision recursions end up 2e-10 from the exact value, 2.11e-10 against 1.96e-10 measured: a ratio of 1.08, the
    largest any run has used -- profiles/r05/parity_allowances.json), and the census is printed: per class the count, the fraction, the worst difference, and cond(EigenMat)
    of the entries beyond rtol against the rest -- the entries beyond rtol are the ill-conditioned ones, on which the
    reference's own LU is rtol ... 1e-3 away from the exact value of its formulas."""
    got, want, labels = np.asarray(got, dtype=float), np.asarray(want, dtype=float), np.asarray(labels)
    fin = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), fin), "%s: finite pattern differs" % what
    rel = np.zeros(got.size)
    rel[fin] = np.abs(got[fin] - want[fin]) / np.abs(want[fin])
    out = {}
    for cls in sorted(set(labels.tolist())):
        sel = fin & (labels == cls)
        bad = np.flatnonzero(sel & (rel > rtol))
        n = int(sel.sum())
        frac = bad.size / max(n, 1)
        cb = [cond_eigenmat(thetas[i], p) for i in bad]
        rest = np.flatnonzero(sel & (rel <= rtol))
        cr = [cond_eigenmat(thetas[i], p) for i in rest[:: max(1, rest.size // 50)]]
        print("%s census [%s]: %d of %d finite entries beyond %.0e of the oracle (%.2f %%, allowed %.2f %%), worst %.2e; "
              "cond(EigenMat) median %.1e for those, %.1e for the rest" % (
                  what, cls, bad.size, n, rtol, 100 * frac, 100 * bounds[cls], rel[bad].max() if bad.size else 0.0,
                  np.median(cb) if cb else 0.0, np.median(cr) if cr else 0.0))
        assert frac <= bounds[cls], "%s [%s]: %.2f %% of the entries beyond %.0e (allowed %.2f %%)" % (what, cls, 100 * frac, rtol, 100 * bounds[cls])
        wd, wo, wratio = 0.0, 0.0, 0.0
        for i in bad:
            truth = arbiter(int(i))
            eg, eo = abs(got[i] - truth), abs(want[i] - truth)
            wd, wo = max(wd, eg / abs(truth)), max(wo, eo / abs(truth))
            if eg > rtol * abs(truth):
                wratio = max(wratio, eg / max(eo, 1e-300))
            assert eg <= max(rtol * abs(truth), arb_factor * eo), (
                "%s [%s]: entry %d is further from the exact value than the oracle (gpu err %.2e, oracle err %.2e)" % (
                    what, cls, i, eg / abs(truth), eo / abs(truth)))
            print("%s [%s]: entry %d: gpu %.2e, oracle %.2e from the exact value, cond(EigenMat) %.1e" % (
                what, cls, i, eg / abs(truth), eo / abs(truth), cond_eigenmat(thetas[i], p)))
        record_allowance("census [%s] (fraction beyond 1e-10 of the oracle; arb_factor %.2f, worst ratio used %.2f)" % (
            cls, arb_factor, wratio), what, bad.size, int(bounds[cls] * n), n, wd, wo, np.median(cb) if cb else None)
        out[cls] = (bad.size, n)
    return out


def assert_same_evaluation(a, b, thetas, p, what="", rtol=1e-8, cond_min=1e5, ceiling=1e-6, thetas_b=None, max_moved_frac=0.02):
    """Two launch shapes of the same evaluation (a, b: log-posteriors of the states `thetas`): within rtol of each other -- the
    bar for WELL-CONDITIONED states, kept as it was --; an entry beyond it must be a flagged ill-conditioned state
    (cond(EigenMat) >= cond_min, where rounding is amplified on both sides) and still within `ceiling`.
    thetas_b: the states b was evaluated at, when the two sides are two sampler kernels' chains.  The kernels round the rank-1
    update of the proposal factor differently and a chain amplifies that from iteration to iteration (a 64-temperature ladder,
    40 iterations: 1.8e-7 in a hot chain's theta, tools/fuzz_sampler.py seed 13) -- where the two states differ by more than
    1e-10 the two log-posteriors are values at DIFFERENT points (each is held to the oracle at its own point elsewhere) and are
    compared through `ceiling` only; at most max_moved_frac of the entries."""
    a, b = np.ravel(np.asarray(a, dtype=float)), np.ravel(np.asarray(b, dtype=float))
    th = np.asarray(thetas, dtype=float).reshape(a.size, -1)
    fin = np.isfinite(a)
    assert np.array_equal(np.isfinite(b), fin), what
    rel = np.zeros(a.size)
    rel[fin] = np.abs(b[fin] - a[fin]) / np.abs(a[fin])
    bad = np.flatnonzero(rel > rtol)
    assert rel.max() <= ceiling, "%s: %.2e apart" % (what, rel.max())
    if thetas_b is not None and bad.size:
        thb = np.asarray(thetas_b, dtype=float).reshape(a.size, -1)
        moved = np.max(np.abs(thb - th) / (1.0 + np.abs(th)), axis=1) > 1e-10
        nm = int(moved[bad].sum())
        assert nm <= max(2, int(max_moved_frac * a.size)), "%s: %d states moved apart between the kernels" % (what, nm)
        if nm:
            print("%s: %d of %d states differ between the two kernels' chains by more than 1e-10 (rounding, amplified): their "
                  "log-posteriors, up to %.1e apart, are values at different points" % (what, nm, a.size, rel[bad][moved[bad]].max()))
        record_allowance("states moved apart between two kernels' chains", what, nm, max(2, int(max_moved_frac * a.size)), a.size)
        bad = bad[~moved[bad]]
    if bad.size:
        record_allowance("kernel pair beyond %.0e on an ill-conditioned state (cond >= %.0e, ceiling %.0e)" % (rtol, cond_min, ceiling),
                         what, bad.size, a.size, a.size, rel[bad].max())
    for i in bad:
        c = cond_eigenmat(th[i], p)
        assert c >= cond_min, "%s: a well-conditioned state (cond(EigenMat) %.1e) differs by %.2e between the kernels" % (what, c, rel[i])
    if bad.size:
        print("%s: %d of %d states beyond %.0e (all ill-conditioned), worst %.2e" % (what, bad.size, a.size, rtol, rel.max()))


def in_zero_root_band(theta, p, q):
    """True when a quadratic factor of theta (AR or MA; log coefficients lq1, lq2 -> q1 = e^lq1, q2 = e^lq2) has two real roots
    with 4 q1 / q2^2 between 2^-54 and 2^-51: the band in which the reference's smaller root -(q2 - sqrt(q2^2 - 4 q1)) / 2 is
    exactly zero or not depending on the last bit of exp() and of the discriminant's rounding -- and with a zero MA root
    the reference's log-density is NaN (carpack.cpp:522-580 divides by it).  Two correct implementations can land on
    different sides; the sampler's unconstrained MA parameters drift through this band."""
    th = np.asarray(theta, dtype=float)
    for lo, m in ((3, p), (3 + p, q)):
        for i in range(m // 2):
            lq1, lq2 = th[lo + 2 * i], th[lo + 2 * i + 1]
            r = np.log2(4.0) + (lq1 - 2.0 * lq2) / np.log(2.0)         # log2(4 q1 / q2^2)
            if -54.0 < r < -51.0:
                return True
    return False


def in_overflow_region(theta, p, q):
    """True when the reference's MA coefficients (carpack.cpp:522-580: the polynomial of the MA roots divided by its
    constant term, i.e. by the product of the roots) exceed 1e150.  CARp::Variance (carpack.cpp:377-409) multiplies two sums
    of them: the products leave the double range, the variance comes out +-inf or NaN depending on the signs of the
    infinities, sigma^2 = ysigma^2 / variance is 0 or NaN, and the log-density the reference returns is an artefact --
    finite values up to 20x away from the exact (quad-precision) one, measured by tools/debug/overflow_probe.py, or NaN.
    The device (which never forms the coefficients: beta(omega_r) = prod (mu_k - omega_r) / mu_k) overflows in the same
    region, to other artefacts.  There is nothing to be identical with: such states are left out of a comparison, counted.
    The MA parameters are unbounded and the likelihood is flat in that direction, so the hottest chains of a long run drift
    there (tools/soak_pt_row.py: 1-2 of 600 states after 1e5 iterations)."""
    if q == 0:
        return False
    import oracle as orc
    with np.errstate(all="ignore"):
        ma = np.abs(orc.ma_coefs(np.asarray(theta, dtype=float), p, q))
    return bool(np.all(np.isfinite(ma)) and ma.max() > 1e150)


def assert_parity_states(got, want, thetas, p, q, rtol=1e-10, what="", arbiter=None, max_overflow_frac=0.01, **kw):
    """assert_parity for sampler states: entries whose FINITE PATTERN differs are excused when the state sits in the
    zero-root band (in_zero_root_band) and the finite side agrees with the exact (quad-precision) value to rtol -- at most
    0.5 % of the entries, each printed; states in the overflow region (in_overflow_region; at most max_overflow_frac) are left out;
    everything else goes through assert_parity."""
    got, want = np.asarray(got, dtype=float), np.asarray(want, dtype=float)
    noise_scale = kw.pop("noise_scale", None)               # (indexed like the arbiter: re-mapped with it below)
    over = np.array([in_overflow_region(th, p, q) for th in thetas], dtype=bool)
    if over.any():
        assert over.sum() <= max(2, int(max_overflow_frac * got.size)), "%s: %d of %d states in the overflow region" % (what, over.sum(), got.size)
        print("%s: %d of %d states sit in the overflow region of the reference's MA coefficients: left out" % (what, over.sum(), got.size))
        record_allowance("overflow region of the reference's MA coefficients (left out)", what, over.sum(),
                         max(2, int(max_overflow_frac * got.size)), got.size)
        keep = np.flatnonzero(~over)
        return assert_parity_states(got[keep], want[keep], np.asarray(thetas)[keep], p, q, rtol, what,
                                    arbiter=(lambda k: arbiter(int(keep[k]))) if arbiter else None,
                                    noise_scale=(lambda k: noise_scale(int(keep[k]))) if noise_scale else None, **kw)
    diff = np.flatnonzero(np.isfinite(got) != np.isfinite(want))
    assert diff.size <= max(2, int(0.005 * got.size)), "%s: %d entries with a different finite pattern" % (what, diff.size)
    keep = np.ones(got.size, dtype=bool)
    if diff.size:
        record_allowance("zero-root band (finite pattern differs)", what, diff.size, max(2, int(0.005 * got.size)), got.size)
    for i in diff:
        assert in_zero_root_band(thetas[i], p, q), "%s: finite pattern differs at %d outside the zero-root band" % (what, i)
        v = got[i] if np.isfinite(got[i]) else want[i]
        truth = arbiter(int(i))
        # (no second implementation to be "no further from the exact value than": where the modal basis is ill conditioned as
        # well, the finite side is held to the forward-error bound of a backward-stable set-up, 4 eps cond(EigenMat) --
        # tools/soak_pt_lane.py met a CARMA(7,6) state with three real roots within 1 % of each other, cond 7e9, on the band:
        # device 2.3e-7 from the exact value, oracle NaN)
        cond = cond_eigenmat(thetas[i], p) if p > 1 else 1.0
        tol = max(rtol, 4.0 * np.finfo(float).eps * cond)
        assert abs(v - truth) <= tol * abs(truth), (what, i, v, truth, cond)
        print("%s: entry %d sits in the zero-root band (device %r, oracle %r, exact %r, cond(EigenMat) %.1e): excused" % (
            what, i, got[i], want[i], truth, cond))
        keep[i] = False
    idx = np.flatnonzero(keep)
    return assert_parity(got[keep], want[keep], rtol, what, arbiter=(lambda k: arbiter(int(idx[k]))) if arbiter else None,
                         noise_scale=(lambda k: noise_scale(int(idx[k]))) if noise_scale else None, **kw)


def queue_get(q, procs, timeout=600.0):
    """q.get() that notices a dead worker at once instead of after the whole time-out."""
    import queue
    import time
    t0 = time.time()
    while True:
        try:
            return q.get(timeout=2.0)
        except queue.Empty:
            dead = [p.exitcode for p in procs if not p.is_alive() and p.exitcode not in (0, None)]
            assert not dead, "worker process exited with code %r" % dead
            assert time.time() - t0 < timeout, "worker timed out"
