#!/usr/bin/env python3
"""Windowed wave pipeline (k_logdens_carma_w, carma_pipew.h) against the oracle and against the one-datum pipeline: parity on
the README fixture / prior-like batches, and launch times over batch sizes.  CARMA_TUNE_WIN_ROWS selects the kernel."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(win_rows, p, q, t, y, e, thetas, ignore_prior, reps=0):
    # a fresh process per setting would be cleaner; the tuning variable is read once per process, so fork
    import subprocess, json, tempfile
    with tempfile.NamedTemporaryFile(suffix=".npz", delete=False) as f:
        np.savez(f, t=t, y=y, e=e, th=thetas)
    env = dict(os.environ, CARMA_TUNE_WIN_ROWS=str(win_rows))
    code = r'''
import sys, json, time, numpy as np
sys.path.insert(0, %r)
import carma_pack_amd as cpa
d = np.load(%r)
ctx = cpa.Context(d["t"], d["y"], d["e"], %d, %d, max_stdev=10.0 * d["y"].std())
th = d["th"]
out = ctx.logdensity(th, ignore_prior=%r)
name = ctx.kernel_name(th.shape[0]) if hasattr(ctx, "kernel_name") else ""
ts = []
nrep = %d
if nrep:
    import torch
    dev = torch.from_numpy(th).cuda()
    o = torch.empty(th.shape[0], dtype=torch.float64, device="cuda")
for _ in range(nrep):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): ctx.logdensity_dev(dev.data_ptr(), th.shape[0], o.data_ptr(), ignore_prior=%r)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 50)
np.save(%r, out)
print(json.dumps(dict(name=name, us=min(ts) * 1e6 if ts else None)))
''' % (ROOT, f.name, p, q, bool(ignore_prior), reps, bool(ignore_prior), f.name + ".out.npy")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    if r.returncode != 0:
        print(r.stdout[-2000:], r.stderr[-4000:])
        raise SystemExit(1)
    info = json.loads(r.stdout.strip().splitlines()[-1])
    return np.load(f.name + ".out.npy"), info


if __name__ == "__main__":
    import oracle as orc
    from carma_pack_amd.synth import theta_batch, prior_like_theta
    g = np.load(os.path.join(ROOT, "tests", "golden", "carma53_readme.npz"))
    t, y, e = g["t"], g["y"], g["yerr"]
    mode = sys.argv[1] if len(sys.argv) > 1 else "parity"
    if mode == "parity":
        orders = ((5, 3), (2, 1), (3, 0), (4, 2), (6, 3), (7, 6))
        if os.environ.get("WIN_ORDERS"):
            orders = [tuple(int(v) for v in x.split(":")) for x in os.environ["WIN_ORDERS"].split(",")]
        for (p, q) in orders:
            rng = np.random.default_rng(100 + p)
            post = theta_batch(rng, 96, p, q, t, y, theta_center=g["theta"][0]) if (p, q) == (5, 3) else None
            prior = np.array([prior_like_theta(rng, p, q, t, y) for _ in range(160)])
            m = orc.OracleModel(t, y, e, p, q, max_stdev=10.0 * y.std())
            for name, th in (("posterior-like", post), ("prior-like", prior)):
                if th is None:
                    continue
                ref = m.logdensity_batch(th, ignore_prior=True)
                got, info = run(4096, p, q, t, y, e, th, True)
                old, _ = run(0, p, q, t, y, e, th, True)
                fin = np.isfinite(ref)
                same = np.array_equal(np.isfinite(got), fin)
                rel = np.abs(got[fin] - ref[fin]) / np.abs(ref[fin]) if same else np.array([np.nan])
                relo = np.abs(old[fin] - ref[fin]) / np.abs(ref[fin])
                print("p=%d q=%d %-14s n=%3d finite %3d pattern %s | window: median %.1e 99%% %.1e max %.1e >1e-10: %d | one-datum: max %.1e >1e-10: %d  [%s]" % (
                    p, q, name, th.shape[0], fin.sum(), same, np.median(rel), np.quantile(rel, 0.99), rel.max(), np.sum(rel > 1e-10),
                    relo.max(), np.sum(relo > 1e-10), info["name"]))
    else:
        p, q = 5, 3
        rng = np.random.default_rng(7)
        for B in [int(x) for x in os.environ.get("WIN_BS", "256,1024,2048,3072,4096,6144,8192,12288,16384").split(",")]:
            th = theta_batch(rng, B, p, q, t, y, theta_center=g["theta"][0])
            _, a = run(1 << 20, p, q, t, y, e, th, False, reps=3)
            _, b = run(0, p, q, t, y, e, th, False, reps=3)
            print("B = %6d  window %8.1f us (%.2e evals/s)   present dispatch %8.1f us (%.2e evals/s)   [%s | %s]" % (
                B, a["us"], B / a["us"] * 1e6, b["us"], B / b["us"] * 1e6, a["name"], b["name"]))
