"""Deterministic tie between the device sampler step and the REFERENCE's arithmetic.

The reference's generator is a time-seeded global (src/random.cpp:20), so its chains cannot be reproduced -- but its STEP
can: oracle/carma_oracle.c `orc_ram_step` is AdaptiveMetro::DoStep (src/steps.cpp:60-107, Accept :36-56, CholUpdateR1
:111-131) and `orc_exchange` is ExchangeStep::DoStep (src/include/steps.hpp:318-362), literal, with the random variates as
INPUTS.  The device tells which variates chain (replica, temperature) uses at iteration i (`carma_pt_debug_draws`: the
kernels' own Philox / Student-t functions, run on the device) and shows its proposal factors (`carma_pt_get_factor`).

Every test walks the device one iteration at a time and, from the device's state BEFORE the iteration, composes the oracle's
steps in the device's order (all RAM steps, then the sweep hot -> cold: SURVEY section 7 sanctions that interleaving):
accept and swap decisions must be identical, theta within 1e-12, the stored log-posterior within 1e-10 of the oracle's
LogDensity (quad-precision arbiter where device and oracle part on an ill-conditioned state), and the factor R within
1e-12 PLUS the band the log-density bar itself implies: the step size of the rank-1 update is sqrt(eta |alpha - 0.25|) with
alpha = exp((l_new - l_old) / T), so a log-density 1e-10 apart moves R by up to ~1e-8 (more next to alpha = 0.25) -- the
oracle step is therefore run three times, with its LogDensity(proposal) as is and moved by -+1e-10 of its magnitude, and
the device's factor must lie within the spread.  A wrong exponent or an off-by-one in eta moves R by percents.  Orders (1,0),
(2,1), (5,3), (7,6); 1, 5, 16 temperatures; the ladder / row / lane kernels and a ladder sharded 3 + 2."""
import os

import numpy as np
import pytest

import oracle as orc

pytestmark = pytest.mark.gpu

RTOL_STATE, RTOL_LP = 1e-12, 1e-10


def _series(n=80, seed=11):
    rng = np.random.default_rng(seed)
    t = np.cumsum(rng.uniform(1.0, 3.0, n))
    y = np.cumsum(rng.standard_normal(n)) * 0.3 + 0.2 * rng.standard_normal(n)
    return t, y - y.mean(), np.full(n, 0.2)


def _sigma0_factor(y, n, d):
    """chol of RunCarmaSampler's initial proposal covariance (src/carmcmc.cpp:132-136; :50-54 for CAR(1))."""
    var = np.mean(y * y) - np.mean(y) ** 2                      # carmcmc.cpp:85-88
    cov = np.eye(d) * 1e-2 * 1e-2
    cov[0, 0] = 2.0 * var * var / n
    cov[2, 2] = var / n
    return np.linalg.cholesky(cov).T                            # arma::chol: upper, Sigma = R^T R (steps.cpp:32)


class _Emu:
    """One iteration of every chain of one block, composed from the oracle's step functions in the device's order."""

    def __init__(self, model, temps, maxiter, truth, noise=None):
        self.m, self.temps, self.maxiter, self.truth, self.narb, self.noise = model, np.asarray(temps), int(maxiter), truth, 0, noise

    def iterate(self, ctx, th, lp, R, it, do_exchange=True):
        """th [Rp][T][d], lp [Rp][T], R [Rp][T][d][d] (copies are returned).  Also returns the decisions taken."""
        th, lp, R = th.copy(), lp.copy(), R.copy()
        Rp, T, d = th.shape
        acc = np.zeros((Rp, T), bool)
        swp = np.zeros((Rp, T), bool)
        band = np.zeros(R.shape)
        tie = np.zeros((Rp, T), bool)                           # u within the bar's band of alpha: either decision is right
        ninf = 0
        for r in range(Rp):
            us = np.zeros(T)
            for i in range(T):
                z, ua, us[i] = ctx.pt_debug_draws(r, i, it)
                a0 = (th[r, i], lp[r, i], R[r, i], z, ua, self.temps[i], it, self.maxiter)
                base = self.m.ram_step(*a0)
                # how far the device's log-density of THIS proposal is from the oracle's (through the batch entry point: the
                # sampler's own evaluation of a rejected proposal is not observable); beyond the bar only on ill-conditioned
                # proposals, where the exact value arbitrates
                width = RTOL_LP
                if np.isfinite(base[4]):
                    thn = th[r, i] + R[r, i].T @ z
                    dl = float(np.ravel(ctx.logdensity(thn))[0])
                    rel = abs(dl - base[4]) / max(1.0, abs(base[4])) if np.isfinite(dl) else 0.0
                    if rel > RTOL_LP:
                        exact = self.truth(thn)
                        eo = abs(base[4] - exact)
                        if abs(dl - exact) > 1.25 * eo + RTOL_LP * max(1.0, abs(exact)) and self.noise is not None:
                            # further than the oracle AT THIS proposal: within the oracle's own error scale around it?  (Where roots
                            # nearly coincide every double-precision implementation's distance from the exact value jumps from one
                            # ulp of theta to the next, helpers.oracle_noise_scale; round 6: the two-sided filter's merge is one more
                            # such implementation -- 8e-8 against the oracle's 3e-8 on one proposal of the (5,3) ladder run)
                            eo = max(eo, 2.0 * self.noise(thn) * abs(exact) / 1.25)
                        assert abs(dl - exact) <= 1.25 * eo + RTOL_LP * max(1.0, abs(exact)), (it, r, i, dl, base[4], exact)
                        self.narb += 1
                        width = 2.0 * rel
                for sh in (width, -width):                       # sensitivity of the factor to the log-density bar
                    a_s, _, _, R_s, _ = self.m.ram_step(*a0, lnew_rel_shift=sh)
                    band[r, i] = np.maximum(band[r, i], np.abs(R_s - base[3]))
                    tie[r, i] |= a_s != base[0]
                acc[r, i], th[r, i], lp[r, i], R[r, i], lnew = base
                ninf += not np.isfinite(lnew)
            if do_exchange:
                for i in range(T - 1, 0, -1):                   # hot -> cold (carmcmc.cpp:147-157)
                    swp[r, i], th[r, i], lp[r, i], th[r, i - 1], lp[r, i - 1] = orc.OracleModel.exchange(
                        th[r, i], lp[r, i], self.temps[i], th[r, i - 1], lp[r, i - 1], self.temps[i - 1], us[i])
        self.band, self.tie = band, tie
        return th, lp, R, acc, swp, ninf


def _close(a, b, rtol, what, it):
    a, b = np.asarray(a), np.asarray(b)
    err = np.abs(a - b) / np.maximum(1.0, np.abs(b))
    assert np.all(err <= rtol), "%s, iteration %d: worst %.3e at %s (device %r, oracle %r)" % (
        what, it, err.max(), np.unravel_index(err.argmax(), err.shape), a.flat[err.argmax()], b.flat[err.argmax()])


def _walk(ctx, emu, niter, t, y, e, p, q, ms, checkpoints=(1, 2, 10), free_run=10):
    """Lock-step comparison over niter iterations; free-running comparison (the emulator on its OWN state) over the first
    free_run.  Returns statistics of what the walk met."""
    from helpers import loglik_truth
    th0, lp0 = ctx.pt_get_chains()
    R0 = ctx.pt_get_factor()
    fth, flp, fR = th0.copy(), lp0.copy(), R0.copy()
    stats = dict(accepted=0, swapped=0, rejected_inf=0, arbitrated=0, steps=0, downdates=0, worst_factor=0.0)
    it0 = ctx.pt_iterations_done()
    for k in range(niter):
        it = it0 + k
        eth, elp, eR, acc, swp, ninf = emu.iterate(ctx, th0, lp0, R0, it)
        if k < free_run:
            fth, flp, fR, _, _, _ = emu.iterate(ctx, fth, flp, fR, it)
        ctx.pt_iterate(1)
        th1, lp1 = ctx.pt_get_chains()
        R1 = ctx.pt_get_factor()
        # decisions: a chain's theta after the iteration is either its old value, its proposal, or a neighbour's -- compare
        # the whole state, which pins accept and swap decisions at once
        assert not emu.tie.any(), "a Metropolis uniform within 1e-10 of alpha (iteration %d): re-seed the test" % it
        _close(th1, eth, RTOL_STATE, "theta", it)
        errR = np.abs(R1 - eR)
        okR = errR <= RTOL_STATE * np.maximum(1.0, np.abs(eR)) + 1.5 * emu.band
        assert okR.all(), "factor, iteration %d: %.3e apart at %s, band %.3e" % (
            it, errR[~okR].max(), np.argwhere(~okR)[0], emu.band[~okR].max())
        stats["worst_factor"] = max(stats["worst_factor"], float((errR / np.maximum(np.abs(eR), 1e-300))[np.abs(eR) > 1e-6].max()))
        bad = np.abs(lp1 - elp) > RTOL_LP * np.maximum(1.0, np.abs(elp))
        for r, i in zip(*np.nonzero(bad)):                        # ill-conditioned state: the exact value arbitrates
            exact = loglik_truth(t, y, e, th1[r, i], p, q)[0]
            assert abs(lp1[r, i] - exact) <= abs(elp[r, i] - exact) * 1.25 + RTOL_LP * max(1.0, abs(exact)), (it, r, i)
            stats["arbitrated"] += 1
        if k < free_run:
            _close(th1, fth, 1e-7, "free-running theta", it)
            _close(R1, fR, 1e-7, "free-running factor", it)
        if it < emu.maxiter:
            stats["downdates"] += int(np.sum(~acc))
        else:
            assert np.array_equal(R1, R0), "factor moved after burn-in (iteration %d)" % it      # niter_ < maxiter_ (steps.cpp:82)
        stats["accepted"] += int(acc.sum())
        stats["swapped"] += int(swp.sum())
        stats["rejected_inf"] += ninf
        stats["steps"] += acc.size
        th0, lp0, R0 = th1, lp1, R1
    return stats


CASES = [  # p, q, T, kernel, iterations, burn-in
    (5, 3, 16, "row", 200, 150),
    (5, 3, 16, "ladder", 60, 40),
    (5, 3, 16, "lane", 60, 40),
    (5, 3, 5, "row", 40, 30),
    (5, 3, 1, "ladder", 40, 30),
    (2, 1, 5, "row", 60, 40),
    (2, 1, 16, "lane", 40, 30),
    (7, 6, 5, "row", 40, 30),
    (7, 6, 16, "ladder", 30, 20),
    (7, 6, 1, "lane", 30, 20),
    (1, 0, 1, "ladder", 60, 40),
    (1, 0, 5, "lane", 40, 30),
]


@pytest.mark.parametrize("p,q,T,kernel,niter,burnin", CASES)
def test_device_step_is_the_reference_step(monkeypatch, p, q, T, kernel, niter, burnin):
    import carma_pack_amd as cpa
    t, y, e = _series()
    ms = 10.0 * y.std()
    monkeypatch.setenv("CARMA_PT_KERNEL", kernel)
    ctx = cpa.Context(t, y, e, p, q, max_stdev=ms)
    Rp = 3
    ctx.pt_create(T, Rp, adapt_iters=burnin, seed=20251003 + 7 * p + T)
    ctx.pt_start(None)
    if kernel != "row" or ctx.pt_kernel() == "row":
        assert ctx.pt_kernel() == kernel
    d = ctx.d
    # Sigma_0 (carmcmc.cpp:132-136) and the ladder T_i = 100^(i / (T - 1)) (carmcmc.cpp:92-95)
    R0 = ctx.pt_get_factor()
    np.testing.assert_allclose(R0, np.broadcast_to(_sigma0_factor(y, t.size, d), R0.shape), rtol=1e-15, atol=0)
    temps = np.exp(np.log(100.0) * np.arange(T) / max(T - 1, 1)) if T > 1 else np.ones(1)
    m = orc.OracleModel(t, y, e, p, q, max_stdev=ms)
    th, lp = ctx.pt_get_chains()
    from helpers import assert_parity, loglik_truth
    flat = th.reshape(-1, d)
    assert_parity(lp.ravel(), m.logdensity_batch(flat), RTOL_LP, "starting log-posterior",
                  arbiter=lambda k: loglik_truth(t, y, e, flat[k], p, q)[0], arb_factor=1.25)
    from helpers import oracle_noise_scale
    emu = _Emu(m, temps, burnin, lambda x: loglik_truth(t, y, e, x, p, q)[0], noise=lambda x: oracle_noise_scale(m, t, y, e, x, p, q))
    stats = _walk(ctx, emu, niter, t, y, e, p, q, ms)
    stats["arbitrated"] += emu.narb
    assert ctx.pt_iterations_done() == niter
    # the walk met every branch of the step: accepts and rejects, up- and downdates, swaps, and (with a ladder) at least
    # one proposal outside the prior during adaptation -- alpha_ = 0 there and the factor is DOWNDATED (steps.cpp:41-46, 95)
    assert 0 < stats["accepted"] < stats["steps"]
    assert stats["downdates"] > 0
    if T > 1:
        assert stats["swapped"] > 0
    if T >= 5 and p > 1:
        assert stats["rejected_inf"] > 0, stats
    assert stats["arbitrated"] <= max(4, stats["steps"] // 50), stats
    from helpers import record_allowance
    record_allowance("sampler_step_arbiter", "test_device_step_is_the_reference_step[%d-%d-%d-%s]" % (p, q, T, kernel),
                     stats["arbitrated"], max(4, stats["steps"] // 50), 2 * stats["steps"], stats["worst_factor"])


def _sharded_worker(q_, blocks):
    import carma_pack_amd as cpa
    from carma_pack_amd import _lib, parallel as par
    P, Q, TG, Rp, NIT, BURN, SEED = 3, 1, sum(blocks), 3, 30, 20, 909
    t, y, e = _series()
    ms = 10.0 * y.std()
    temps = par.ladder_temperatures(TG)
    comm = _lib.Comm(_lib.Comm.unique_id(), 1, 0, device=0) if len(blocks) > 1 else None
    ctxs, slot0 = [], 0
    for Tl in blocks:
        c = cpa.Context(t, y, e, P, Q, max_stdev=ms)
        c.pt_create(Tl, Rp, BURN, seed=SEED, temperatures=temps[slot0:slot0 + Tl])
        c.pt_shard(TG, slot0, 0)
        c.pt_start(None)
        ctxs.append(c)
        slot0 += Tl
    m = orc.OracleModel(t, y, e, P, Q, max_stdev=ms)

    class Both:                                              # the blocks seen as one ladder
        d = ctxs[0].d

        def pt_debug_draws(self, r, i, it):
            for c, Tl, s0 in zip(ctxs, blocks, np.cumsum([0] + list(blocks[:-1]))):
                if s0 <= i < s0 + Tl:
                    return c.pt_debug_draws(r, i - s0, it)

        def pt_get_chains(self):
            a = [c.pt_get_chains() for c in ctxs]
            return np.concatenate([x[0] for x in a], axis=1), np.concatenate([x[1] for x in a], axis=1)

        def pt_get_factor(self):
            return np.concatenate([c.pt_get_factor() for c in ctxs], axis=1)

        def pt_iterate(self, k):
            _lib.pt_iterate_sharded(ctxs, k, comm)

        def pt_iterations_done(self):
            return ctxs[0].pt_iterations_done()

        def logdensity(self, x):
            return ctxs[0].logdensity(x)

    try:
        from helpers import loglik_truth
        from helpers import oracle_noise_scale
        stats = _walk(Both(), _Emu(m, temps, BURN, lambda x: loglik_truth(t, y, e, x, P, Q)[0],
                                   noise=lambda x: oracle_noise_scale(m, t, y, e, x, P, Q)), NIT, t, y, e, P, Q, ms)
        q_.put(stats)
    except BaseException as ex:                              # the assertion text must reach the parent
        q_.put(repr(ex))
    if comm is not None:
        comm.close()


def test_sharded_ladder_step_is_the_reference_step():
    """The same lock-step comparison with the ladder cut 3 + 2 (carma_pt_iterate_sharded; boundary rows through RCCL send/recv
    to the process's own rank): RAM steps of both blocks, the in-block sweeps and the boundary swap, composed hot -> cold, are
    the oracle's steps on the device's variates."""
    import torch.multiprocessing as mp
    from helpers import queue_get
    ctx = mp.get_context("spawn")
    q_ = ctx.Queue()
    p_ = ctx.Process(target=_sharded_worker, args=(q_, [3, 2]))
    p_.start()
    stats = queue_get(q_, [p_], 600)
    p_.join(120)
    assert isinstance(stats, dict), stats
    assert p_.exitcode == 0
    assert stats["swapped"] > 0 and 0 < stats["accepted"] < stats["steps"]
