"""Evaluate the states a soak run flagged (gpurun_out/soak_fail_*.npz, copied to tools/debug/data/) with every log-density kernel."""
import glob, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import carma_pack_amd as cpa
for f in sorted(glob.glob(os.path.join(ROOT, "tools/debug/data/soak_fail_*.npz"))):
    z = np.load(f)
    p, q = [int(s[1:]) for s in os.path.basename(f).split("_")[2:4]]
    t, y, yerr, th, got, want = z["t"], z["y"], z["yerr"], z["theta"], z["got"], z["want"]
    odd = np.flatnonzero(np.isfinite(got) != np.isfinite(want))
    ctx = cpa.Context(t, y, yerr, p, q, max_stdev=float(z["max_stdev"]))
    for shape in (None, "p3l", "grp", "lane"):
        if shape: os.environ["CARMA_LOGDENS_SHAPE"] = shape
        try:
            lp = ctx.logdensity(th)
            print(os.path.basename(f), shape, "states", odd, "sampler lp", got[odd], "kernel lp", lp[odd], "oracle", want[odd],
                  "| others agree with sampler lp to", np.nanmax(np.abs(np.delete(lp, odd) - np.delete(got, odd)) / np.abs(np.delete(got, odd))))
        except Exception as e:
            print(shape, "ERR", e)
