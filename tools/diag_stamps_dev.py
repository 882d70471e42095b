"""As tools/diag_stamps.py, but on the PRODUCT's path: parameter vectors resident in HBM, launches back to back through the
_dev entry point (the host path of diag_stamps.py copies theta in front of every launch -- its first touch in the kernel then goes to
HBM).  Prints the diagnostic build's per-phase marks of the last launches."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import carma_pack_amd._lib as L0
diag = os.environ.get("CARMA_DIAG_LIB") or os.path.join(ROOT, "build_diag", "libcarma_mi355_diag.so")
L0.LIB_PATH = diag
L0.lib = L0._load()
from carma_pack_amd.synth import theta_batch
g = np.load(os.path.join(ROOT, 'tests/golden/carma53_readme.npz'))
t, y, yerr = g['t'], g['y'], g['yerr']
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx = L0.Context(t, y, yerr, 5, 3, max_stdev=10 * np.sqrt(np.mean(y * y) - np.mean(y) ** 2))
pool = [torch.from_numpy(theta_batch(np.random.default_rng(2 + i), B, 5, 3, t, y, theta_center=g['theta'][0])).cuda() for i in range(4)]
out = torch.empty(B, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
for i in range(12):
    ctx.logdensity_dev(pool[i % 4].data_ptr(), B, out.data_ptr())
torch.cuda.synchronize()
print("finite", int(torch.isfinite(out).sum()))
