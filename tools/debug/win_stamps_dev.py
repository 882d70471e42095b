import os, sys, numpy as np, torch
ROOT = os.getcwd(); sys.path.insert(0, ROOT)
os.environ["CARMA_LIB_PATH"] = os.path.join(ROOT, "build_var", "wstamps.so")
import carma_pack_amd as cpa
from carma_pack_amd.synth import theta_batch
g = np.load(os.path.join(ROOT, "tests", "golden", "carma53_readme.npz")); t, y, e = g["t"], g["y"], g["yerr"]
B = int(sys.argv[1])
ctx = cpa.Context(t, y, e, 5, 3, max_stdev=10.0 * y.std())
th = torch.from_numpy(theta_batch(np.random.default_rng(2), B, 5, 3, t, y, theta_center=g["theta"][0])).cuda()
o = torch.empty(B, dtype=torch.float64, device="cuda")
for _ in range(6): ctx.logdensity_dev(th.data_ptr(), B, o.data_ptr())
torch.cuda.synchronize()
