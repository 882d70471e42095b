"""Fuzz of the sampler launcher: random orders, ladder lengths and replica counts AROUND THE GRID SIZES at which the row
kernel changes its build (one / two / three workgroups per CU) or hands over to the ladder kernel.  Per case the two
kernels (CARMA_PT_KERNEL=row / ladder, separate processes: the choice is read once) run the same seed and start; their
chains must agree (accept / swap decisions identical, values to rounding) and the stored log-posteriors must be the
oracle's LogDensity of the chain states.  Run on the GPU box:  python tools/fuzz_sampler.py [cases] [seed]"""
import os, subprocess, sys, json, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def cases(ncase, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(ncase):
        p = int(rng.integers(1, 8))                                  # (1: CAR(1), on the large-ensemble path since round 4)
        q = int(rng.integers(0, p)) if p > 1 else 0
        T = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 12, 16, 17, 20, 33, 64, 65, 70]))
        wg = (T + 3) // 4
        rows = int(rng.choice([1, 2, 255, 256, 257, 511, 512, 513, 767, 768, 769, 1000])) if rng.random() < 0.6 else int(rng.integers(1, 900))
        R = max(1, rows // wg + int(rng.integers(-1, 2)))
        n = int(rng.choice([9, 31, 32, 33, 45, 47, 48, 49, 77, 80, 100]))        # (below 32 data the ladder kernel takes over)
        out.append((p, q, T, R, n, int(rng.integers(1, 10 ** 6))))
    return out


def worker(kern, spec_file, out_file):
    os.environ["CARMA_PT_KERNEL"] = kern
    import carma_pack_amd as cpa
    from helpers import irregular_series
    res = []
    for (p, q, T, R, n, seed) in json.load(open(spec_file)):
        t, y, yerr = irregular_series(n, seed=seed)
        ctx = cpa.Context(t, y, yerr, p, q, max_stdev=10.0 * y.std())
        try:
            ctx.pt_create(T, R, adapt_iters=25, seed=seed)
        except ValueError as ex:                                      # a ladder too long for one workgroup of the fall-back kernel
            res.append(dict(kernel="rejected: " + str(ex)[-60:]))
            continue
        ctx.pt_start(None)
        ctx.pt_iterate(40)
        th, lp = ctx.pt_get_chains()
        acc, swp = ctx.pt_stats()
        res.append(dict(kernel=ctx.pt_kernel(), th=th, lp=lp, acc=acc, swp=swp))
    np.save(out_file, np.array(res, dtype=object), allow_pickle=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker(sys.argv[2], sys.argv[3], sys.argv[4])
        sys.exit(0)
    ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    cs = cases(ncase, seed)
    tmp = tempfile.mkdtemp()
    spec = os.path.join(tmp, "spec.json")
    json.dump(cs, open(spec, "w"))
    outs = {}
    KA = os.environ.get("FUZZ_KERNEL", "row")               # "lane": the large-ensemble path (carma_pt_lane.hip) against the ladder kernel
    for kern in (KA, "ladder"):
        f = os.path.join(tmp, kern + ".npy")
        subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", kern, spec, f], check=True)
        outs[kern] = np.load(f, allow_pickle=True)
    import oracle as orc
    from helpers import assert_parity_states, irregular_series, loglik_truth
    fails, kinds = 0, {}
    for c, a, b in zip(cs, outs[KA], outs["ladder"]):
        p, q, T, R, n, sd = c
        kinds[a["kernel"][:8]] = kinds.get(a["kernel"][:8], 0) + 1
        if a["kernel"].startswith("rejected") or b["kernel"].startswith("rejected"):
            if a["kernel"] != b["kernel"]:
                fails += 1
                print("FAILED CARMA(%d,%d) T=%d R=%d: %s / %s" % (p, q, T, R, a["kernel"], b["kernel"]))
            continue
        try:
            assert b["kernel"] == "ladder"
            # (values to rounding: the kernels round the rank-1 update differently and a chain amplifies that from iteration to
            # iteration -- 1 component in 3.4 million at 1.2e-6 after 40 iterations, fuzz_sampler_lane_v1.txt; the decisions below
            # are compared exactly)
            np.testing.assert_allclose(a["th"], b["th"], rtol=1e-5, atol=1e-8)
            fin = np.isfinite(b["lp"])
            assert np.array_equal(np.isfinite(a["lp"]), fin)
            from helpers import assert_same_evaluation
            assert_same_evaluation(b["lp"], a["lp"], b["th"], p, "states", thetas_b=a["th"])
            assert np.array_equal(a["acc"], b["acc"]) and np.array_equal(a["swp"], b["swp"])
            t, y, yerr = irregular_series(n, seed=sd)
            m = orc.OracleModel(t, y, yerr, p, q, max_stdev=10.0 * y.std())
            flat = a["th"].reshape(-1, 3 + p + q)
            sel = np.random.default_rng(sd).choice(flat.shape[0], size=min(200, flat.shape[0]), replace=False)
            assert_parity_states(a["lp"].reshape(-1)[sel], m.logdensity_batch(flat[sel], nthreads=8), flat[sel], p, q, 1e-10, "states",
                                 arbiter=lambda i: loglik_truth(t, y, yerr, flat[sel][i], p, q)[0], max_arb_frac=0.1, arb_factor=8.0,
                                 max_overflow_frac=0.05)
        except AssertionError as ex:
            fails += 1
            print("FAILED CARMA(%d,%d) T=%d R=%d n=%d seed=%d (%s): %s" % (p, q, T, R, n, sd, a["kernel"], str(ex)[:300]), flush=True)
    if os.environ.get("FUZZ_VERBOSE"):
        for c, a in zip(cs, outs[KA]):
            print("CARMA(%d,%d) T=%2d R=%3d workgroups %4d n=%2d -> %s" % (c[0], c[1], c[2], c[3], c[3] * ((c[2] + 3) // 4), c[4], a["kernel"][:8]))
    print("%d cases, %d failed; kernel the '%s' run was on: %s" % (len(cs), fails, KA, kinds))
