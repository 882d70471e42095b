"""CarmaSample post-processing on the device (carma_post.hip): time of sigma per sample and of the PSD band (1000 frequencies,
three percentiles) for sample counts up to all 3.2 million cold-chain samples of BASELINE configs[2], against the numpy
restatement (oracle/post.py) on a small count.  Run under rocprofv3 --kernel-trace --stats for the per-kernel times."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import carma_pack_amd as cpa
from carma_pack_amd import _lib, carma_pack as cp
from carma_pack_amd.synth import theta_batch
import oracle as orc

g = np.load(os.path.join(ROOT, "tests", "golden", "carma53_readme.npz"))
t, y, yerr = g["t"], g["y"], g["yerr"]
rng = np.random.default_rng(3)
NS = [int(x) for x in os.environ.get("POST_PROBE_NS", "1000,75000,400000,3200000").split(",")]
freq = np.exp(np.linspace(np.log(1.0 / (t.max() - t.min())), np.log(0.5 / np.diff(t).min()), 1000))
for ns in NS:
    base = theta_batch(rng, min(ns, 4096), 5, 3, t, y, theta_center=g["theta"][0])
    base = base[:: 2]                                      # posterior-like half of the generator's output
    th = base[rng.integers(0, base.shape[0], ns)] + 1e-3 * rng.standard_normal((ns, 11))
    roots = cp._roots_from_log_quads(th[:, 3:8])
    ar = cp._poly_from_roots(roots).real
    c = cp._poly_from_roots(cp._roots_from_log_quads(th[:, 8:11]))
    ma = (c / c[:, 3:4])[:, ::-1].real
    t0 = time.perf_counter(); sig = _lib.sigma_noise_batch(roots, ma, th[:, 0] ** 2); t_sig = time.perf_counter() - t0
    ok = np.isfinite(sig)
    sig[~ok] = 1.0
    _lib.psd_band(ar[:256], ma[:256], sig[:256], freq, [16.0, 50.0, 84.0])
    t0 = time.perf_counter(); band = _lib.psd_band(ar, ma, sig, freq, [16.0, 50.0, 84.0]); t_band = time.perf_counter() - t0
    line = "ns = %8d: sigma %.4f s, PSD band (1000 frequencies x %d samples = %.2e values) %.3f s (%.1f GB of grid)" % (
        ns, t_sig, ns, 1000.0 * ns, t_band, 8e-9 * 1000 * ns)
    if ns <= 75000:
        t0 = time.perf_counter(); want = orc.post.psd_band(ar, ma, sig, freq, [16.0, 50.0, 84.0]); t_np = time.perf_counter() - t0
        line += " | numpy restatement %.2f s, max rel diff %.1e" % (t_np, np.nanmax(np.abs(band - want) / np.abs(want)))
    print(line, flush=True)
