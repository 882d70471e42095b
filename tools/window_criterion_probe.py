#!/usr/bin/env python3
"""Round 6: where does SERIES_WINDOW_OK (carma_capi.hip: at most 10 % of the chunk spans longer than the shortest window the prior
admits) draw the line too early?  OGLE-LMC-LPV-00007 (the quick-start series: n = 437, seasons, min dt 1 d, median 3 d) fails it at
every order (20 ... 55 % of the spans), BASELINE configs[3]'s series at ~100 %, the README series passes at 8 %.  For each order:
the log-density launch (prior-like batch) and a single ladder of 10 temperatures (the quick-start call's shape) and 16 x 64 ladders,
on the one-datum pipeline and with the two-sided window pipeline forced; chains prior-like and 2000 iterations old."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import carma_pack_amd as cpa
from carma_pack_amd.synth import theta_batch
from carma_pack_amd import parallel as par

g = np.load(os.path.join(ROOT, "tests", "golden", "carma53_readme.npz"))


def series(name):
    if name == "ogle":
        d = np.loadtxt(os.path.join(ROOT, "tests", "golden", "ogle_lmc_lpv_00007.dat"))
        return d[:, 0], d[:, 1], d[:, 2]
    if name.startswith("config3_"):                     # BASELINE configs[3]'s time steps (0.1 + |Cauchy|), the first n data
        from carma_pack_amd.synth import config4_series
        t4, y4, e4, _ = config4_series(int(name.split("_")[1]), seed=4)
        return t4, y4, e4
    if name == "close_pair":                            # the README series with ONE more datum 0.01 after another: max_freq x 100
        t, y, e = g["t"], g["y"], g["yerr"]
        k = 100
        return np.insert(t, k + 1, t[k] + 0.01), np.insert(y, k + 1, y[k]), np.insert(e, k + 1, e[k])
    raise SystemExit(name)


names = [a for a in sys.argv[1:] if "," not in a] or ["ogle"]
orders = [(int(a), int(b)) for a, b in (s.split(",") for s in sys.argv[1:] if "," in s)] or [(6, 0), (5, 3), (7, 6), (3, 2), (2, 1)]
for name in names:
  t, y, e = series(name)
  ts = np.sort(t)
  wmin = 0.5 * 600.0 * np.diff(ts).min() / (2 * np.pi)
  print(json.dumps(dict(series=name, n=len(t), min_dt=float(np.diff(ts).min()), median_dt=float(np.median(np.diff(ts))), shortest_window=float(wmin),
                        spans_over={p: round(float(((ts[15 - p:] - ts[:len(ts) - 15 + p]) > wmin).mean()), 3) for p in range(2, 8)})), flush=True)
  for p, q in orders:
      ctx = cpa.Context(t, y, e, p, q)
      for B in (256, 1024):
          th = theta_batch(np.random.default_rng(3), B, p, q, t, y)
          dev = torch.from_numpy(th).cuda()
          o = torch.empty(B, dtype=torch.float64, device="cuda")
          for forced in (0, 1):
              cpa._lib.tune_reset()
              if forced:
                  cpa._lib.tune_set("WIN_ROWS", 1 << 20)
              for _ in range(30):
                  ctx.logdensity_dev(dev.data_ptr(), B, o.data_ptr())
              best = 1e9
              for _ in range(5):
                  torch.cuda.synchronize()
                  t0 = time.perf_counter()
                  for _ in range(200):
                      ctx.logdensity_dev(dev.data_ptr(), B, o.data_ptr())
                  torch.cuda.synchronize()
                  best = min(best, (time.perf_counter() - t0) / 200)
              print(json.dumps(dict(series=name, p=p, q=q, leg="logdensity", B=B, forced=forced, us=round(best * 1e6, 2), kernel=ctx.kernel_name(B))), flush=True)
      cpa._lib.tune_reset()
      del ctx
      for T, R in ((10, 1), (16, 64)):
          for warm in (0, 2000):
              for two in (0, 2):
                  ctx = cpa.Context(t, y, e, p, q)
                  cpa._lib.tune_set("PT_ROW_WIN", two)
                  ctx.pt_create(T, R, 10 ** 9, seed=5)
                  ctx.pt_start(None)
                  if warm:
                      cpa._lib.tune_set("PT_ROW_WIN", 0)
                      ctx.pt_iterate(warm)
                      cpa._lib.tune_set("PT_ROW_WIN", two)
                  ctx.pt_iterate(50)
                  t0 = time.perf_counter()
                  ctx.pt_iterate(1000)
                  dt = time.perf_counter() - t0
                  print(json.dumps(dict(series=name, p=p, q=q, leg="sampler %d x %d" % (T, R), chains="prior-like" if not warm else "%d iterations old" % warm,
                                        pipeline="two-sided (forced)" if two else "one-datum", us_per_iteration=round(1e6 * dt / 1000, 2), kernel=ctx.pt_kernel())), flush=True)
                  del ctx
      cpa._lib.tune_reset()
