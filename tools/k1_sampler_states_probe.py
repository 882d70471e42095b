"""Why is the log-density launch inside the large-ensemble sampler slower (256 us) than the throughput leg's (215-220 us)?
Times k_logdens_carma_lane<5> at 65 536 evaluations on (a) bench.py's tiled pool, (b) the states of a 16 x 4096 run after
N iterations in the sampler's lane order, (c) the same states sorted so that the chains with a real root pair share waves --
and counts the waves that hold at least one such chain (they take the two-exponentials-per-pair path, lane_any in
carma_lane.h).  Run on the GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import carma_pack_amd as cpa
from carma_pack_amd.synth import theta_batch

g = np.load(os.path.join(ROOT, "tests", "golden", "carma53_readme.npz"))
t, y, yerr = g["t"], g["y"], g["yerr"]
p, q, BT = 5, 3, 65536
ms = 10 * np.sqrt(np.mean(y * y) - np.mean(y) ** 2)
ctx = cpa.Context(t, y, yerr, p, q, max_stdev=ms)
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream()


def timeit(th, label):
    d = torch.from_numpy(np.ascontiguousarray(th)).to(dev)
    out = torch.empty(BT, dtype=torch.float64, device=dev)
    for _ in range(3):
        ctx.logdensity_dev(d.data_ptr(), BT, out.data_ptr(), ignore_prior=False, stream=stream.cuda_stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(20):
        ctx.logdensity_dev(d.data_ptr(), BT, out.data_ptr(), ignore_prior=False, stream=stream.cuda_stream)
    e1.record(stream)
    torch.cuda.synchronize()
    # a quadratic factor x^2 + b x + c (theta = log c, log b) has real roots when b^2 > 4 c
    c0, b0 = np.exp(th[:, 3]), np.exp(th[:, 4])
    c1, b1 = np.exp(th[:, 5]), np.exp(th[:, 6])
    real = (b0 * b0 > 4 * c0) | (b1 * b1 > 4 * c1)
    waves = real.reshape(-1, 64).any(axis=1)
    r0, r1 = (b0 * b0 > 4 * c0), (b1 * b1 > 4 * c1)
    w0, w1 = r0.reshape(-1, 64).any(axis=1), r1.reshape(-1, 64).any(axis=1)

    fin = np.isfinite(out.cpu().numpy())
    print("%-58s %7.1f us per launch | chains with a real pair %5.1f %%, waves holding one %5.1f %% | finite %5.1f %%" % (
        label, 1e3 * e0.elapsed_time(e1) / 20, 100 * real.mean(), 100 * waves.mean(), 100 * fin.mean()), flush=True)
    print("      pair 0 real in %.1f %% of the chains (%.1f %% of the waves), pair 1 in %.1f %% (%.1f %%)" % (
        100 * r0.mean(), 100 * w0.mean(), 100 * r1.mean(), 100 * w1.mean()))
    return real


rng = np.random.default_rng(2)
pool = theta_batch(rng, 1024, p, q, t, y, theta_center=g["theta"][0])
timeit(np.tile(pool, (BT // 1024, 1)), "bench.py's pool (1024 vectors tiled)")
timeit(np.tile(pool[::2], (BT // 512, 1)), "  its posterior-like half")
timeit(np.tile(pool[1::2], (BT // 512, 1)), "  its prior-like half")
ctx.pt_create(16, 4096, adapt_iters=10 ** 9, seed=3)
ctx.pt_start(None)
for it in (0, 30, 200):
    if it:
        ctx.pt_iterate(it)
    th, lp = ctx.pt_get_chains()
    flat = th.reshape(-1, th.shape[-1])
    real = timeit(flat, "sampler states after %3d iterations, lane order" % it)
    order = np.argsort(~real, kind="stable")
    timeit(flat[order], "  the same, chains with a real pair packed together")
    blk = np.arange(flat.shape[0]) // 256
    order_wg = np.lexsort((np.arange(flat.shape[0]), ~real, blk))
    timeit(flat[order_wg], "  the same, packed inside blocks of 256 (4 waves)")
    half = np.concatenate([flat[order][: BT // 2], flat[order][: BT // 2]])
    timeit(half, "  the flagged-first half of the packed batch, twice")
    tail = np.concatenate([flat[order][BT // 2:], flat[order][BT // 2:]])
    timeit(tail, "  the unflagged half of the packed batch, twice")
    for k in (0, 8, 15):
        sub = th[:, k, :]
        c0, b0, c1, b1 = np.exp(sub[:, 3]), np.exp(sub[:, 4]), np.exp(sub[:, 5]), np.exp(sub[:, 6])
        print("      temperature %2d: %5.1f %% of the chains have a real pair" % (k, 100 * ((b0 * b0 > 4 * c0) | (b1 * b1 > 4 * c1)).mean()))
