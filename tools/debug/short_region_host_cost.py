"""Where the host side of a 20-step timed region goes: event records, launches, the final synchronize."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import carma_pack_amd as cpa
from carma_pack_amd.synth import theta_batch
g = np.load(os.path.join(ROOT, "tests/golden/carma53_readme.npz"))
t, y, yerr = g["t"], g["y"], g["yerr"]
ctx = cpa.Context(t, y, yerr, 5, 3, max_stdev=10 * y.std())
dev = torch.device("cuda")
B = 1024
th = torch.from_numpy(theta_batch(np.random.default_rng(2), B, 5, 3, t, y, theta_center=g["theta"][0])).to(dev)
out = torch.empty(B, dtype=torch.float64, device=dev)
stream = torch.cuda.current_stream(); sh = stream.cuda_stream
for _ in range(3000): ctx.logdensity_dev(th.data_ptr(), B, out.data_ptr(), stream=sh)
torch.cuda.synchronize()
for mode in ("sync", "spin"):
    rows = []
    for rep in range(30):
        for _ in range(5): ctx.logdensity_dev(th.data_ptr(), B, out.data_ptr(), stream=sh)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); ev0.record(stream)
        t1 = time.perf_counter()
        for _ in range(20): ctx.logdensity_dev(th.data_ptr(), B, out.data_ptr(), stream=sh)
        t2 = time.perf_counter(); ev1.record(stream)
        t3 = time.perf_counter()
        if mode == "spin":
            while not ev1.query(): pass
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        rows.append(((t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, (t4 - t3) * 1e6, (t4 - t0) * 1e6, ev0.elapsed_time(ev1) * 1e3))
    r = np.median(np.array(rows), axis=0)
    print("%s: record0 %.1f us, 20 launches %.1f us, record1 %.1f us, wait %.1f us, total %.1f us; device (events) %.1f us -> host-only %.1f us" % (
        mode, r[0], r[1], r[2], r[3], r[4], r[5], r[4] - r[5]))
