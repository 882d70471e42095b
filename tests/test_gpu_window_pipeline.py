"""The windowed wave pipeline (carma_pipew.h, k_logdens_carma_w<P>: round 5's blocked form of kfilter.cpp:189-215) and its TWO-SIDED
form (k_logdens_carma_w2<P>, round 6: forward over the first half of the series, backward over the second, merged at the meeting
time -- the default up to four evaluations per CU for series whose sampling suits it), each FORCED on every launch size through
CARMA_TUNE_WIN_ROWS / CARMA_TUNE_WIN2_EVALS, against the oracle:
every order, posterior-like and prior-like parameters (re-base data open chunks, rows of a workgroup end at different chunk
counts), a series long enough for hundreds of chunks, the prior's -inf pattern, and a series on which the dispatch itself would
NOT take it (SERIES_WINDOW_OK, carma_types.h).  The tuning variable is part of the child's environment."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

import oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, json, numpy as np
sys.path.insert(0, %r)
import carma_pack_amd as cpa
d = np.load(sys.argv[1])
ctx = cpa.Context(d["t"], d["y"], d["e"], int(d["p"]), int(d["q"]), max_stdev=float(d["ms"]))
th = d["th"]
out = ctx.logdensity(th, ignore_prior=bool(d["ip"]))
np.save(sys.argv[1] + ".out.npy", out)
print(json.dumps(dict(name=ctx.kernel_name(th.shape[0]))))
'''


def _window(t, y, e, p, q, ms, th, ignore_prior, two_sided=False):
    with tempfile.TemporaryDirectory() as tmp:
        f = os.path.join(tmp, "in.npz")
        np.savez(f, t=t, y=y, e=e, p=p, q=q, ms=ms, th=th, ip=ignore_prior)
        r = subprocess.run([sys.executable, "-c", CHILD % ROOT, f],
                           env=dict(os.environ, CARMA_TUNE_WIN_ROWS="4096", CARMA_TUNE_WIN2_EVALS="1000000" if two_sided else "0"),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        assert json.loads(r.stdout.strip().splitlines()[-1])["name"] == ("k_logdens_carma_w2<%d>" if two_sided else "k_logdens_carma_w<%d>") % p
        return np.load(f + ".out.npy")


SIDES = pytest.mark.parametrize("two_sided", [False, True], ids=["one-sided", "two-sided"])


@SIDES
@pytest.mark.parametrize("p,q", [(2, 1), (3, 0), (4, 3), (5, 3), (6, 2), (7, 6)])
def test_window_pipeline_vs_oracle(golden_dir, p, q, two_sided):
    from helpers import assert_parity, loglik_truth, prior_like_theta, theta_batch
    g = np.load(os.path.join(golden_dir, "carma53_readme.npz"))
    t, y, e = g["t"], g["y"], g["yerr"]
    ms = 10.0 * y.std()
    rng = np.random.default_rng(500 + 10 * p + q)
    th = np.array([prior_like_theta(rng, p, q, t, y) for _ in range(101)])
    if (p, q) == (5, 3):
        th = np.concatenate([th, theta_batch(rng, 64, p, q, t, y, theta_center=g["theta"][0], frac_post=1.0)])
    m = orc.OracleModel(t, y, e, p, q, max_stdev=ms)
    for ip in (True, False):
        got = _window(t, y, e, p, q, ms, th, ip, two_sided)
        assert_parity(got, m.logdensity_batch(th, ignore_prior=ip), 1e-10, "window pipeline (%d,%d) ignore_prior=%d two_sided=%d" % (p, q, ip, two_sided),
                      arbiter=lambda k: loglik_truth(t, y, e, th[k], p, q)[0] if not ip else
                      loglik_truth(t, y, e, th[k], p, q)[0], arb_factor=1.25, max_arb_frac=0.04)


@SIDES
def test_window_pipeline_long_series(two_sided):
    from helpers import assert_parity, irregular_series, loglik_truth, prior_like_theta
    t, y, e = irregular_series(3000, seed=12)
    ms = 10.0 * y.std()
    rng = np.random.default_rng(8)
    th = np.array([prior_like_theta(rng, 5, 2, t, y) for _ in range(37)])
    m = orc.OracleModel(t, y, e, 5, 2, max_stdev=ms)
    got = _window(t, y, e, 5, 2, ms, th, True, two_sided)
    assert_parity(got, m.logdensity_batch(th, ignore_prior=True), 1e-10, "window pipeline, n = 3000, two_sided=%d" % two_sided,
                  arbiter=lambda k: loglik_truth(t, y, e, th[k], 5, 2)[0], arb_factor=1.25, max_arb_frac=0.06)


def test_series_window_criterion():
    """SERIES_WINDOW2_OK / _SMALL (set at context creation, carma_types.h: the chunks a row with the SHORTEST window the prior admits
    needs, against ceil(n / (16 - p)): <= 2 / <= 3.5).  The README series (1.0-1.08) and the OGLE quick-start series (1.14-1.41:
    seasons) have it: the two-sided window pipeline up to six evaluations per CU.  BASELINE configs[3]'s time steps (0.1 + |Cauchy|:
    2.5 at p = 7) have the SMALL form only -- two-sided with a CU per workgroup, the one-datum pipeline beyond --, one close pair of
    data in the README series (max_freq x 100: 9-13) has neither; and the window pipelines, forced, are still right on such series
    (most of their chunks are cut short there: slow, not wrong)."""
    import carma_pack_amd as cpa
    from carma_pack_amd.synth import config4_series
    from helpers import assert_parity, loglik_truth, prior_like_theta
    if os.environ.get("CARMA_TUNE_WIN_ROWS") is not None or os.environ.get("CARMA_TUNE_WIN2_EVALS") is not None:
        pytest.skip("the dispatch under test is overridden by CARMA_TUNE_WIN_ROWS / CARMA_TUNE_WIN2_EVALS")
    g = np.load(os.path.join(ROOT, "tests", "golden", "carma53_readme.npz"))
    ctx = cpa.Context(g["t"], g["y"], g["yerr"], 5, 3)
    assert ctx.kernel_name(1024) == "k_logdens_carma_w2<5>" and ctx.kernel_name(1536) == "k_logdens_carma_w2<5>"
    assert ctx.kernel_name(1537) == "k_logdens_carma_p3l<5>"
    t, y, e, _ = config4_series(1500, seed=4)
    ms = 10.0 * y.std()
    c4 = cpa.Context(t, y, e, 7, 6, max_stdev=ms)
    assert c4.kernel_name(64) == c4.kernel_name(512) == "k_logdens_carma_w2<7>" and c4.kernel_name(513) == c4.kernel_name(1024) == "k_logdens_carma_p3l<7>"
    d = np.loadtxt(os.path.join(ROOT, "tests", "golden", "ogle_lmc_lpv_00007.dat"))
    for p, q in ((2, 1), (6, 0), (7, 6)):
        co = cpa.Context(d[:, 0], d[:, 1], d[:, 2], p, q)
        assert co.kernel_name(64) == co.kernel_name(1536) == "k_logdens_carma_w2<%d>" % p, (p, co.kernel_name(64))
    k = 100
    cc = cpa.Context(np.insert(g["t"], k + 1, g["t"][k] + 0.01), np.insert(g["y"], k + 1, g["y"][k]), np.insert(g["yerr"], k + 1, g["yerr"][k]), 5, 3)
    assert cc.kernel_name(64) == cc.kernel_name(1024) == "k_logdens_carma_p3l<5>"
    rng = np.random.default_rng(77)
    th = np.array([prior_like_theta(rng, 7, 6, t, y) for _ in range(40)])
    m = orc.OracleModel(t, y, e, 7, 6, max_stdev=ms)
    want = m.logdensity_batch(th, ignore_prior=True)
    arb = lambda k: loglik_truth(t, y, e, th[k], 7, 6)[0]   # noqa: E731
    assert_parity(c4.logdensity(th, ignore_prior=True), want, 1e-10, "configs[3]-like series, dispatch", arbiter=arb, arb_factor=1.25,
                  max_arb_frac=0.1)
    for two_sided in (False, True):
        assert_parity(_window(t, y, e, 7, 6, ms, th, True, two_sided), want, 1e-10,
                      "configs[3]-like series, window pipeline forced, two_sided=%d" % two_sided, arbiter=arb, arb_factor=1.25, max_arb_frac=0.1)
