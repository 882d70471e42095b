#!/usr/bin/env python3
"""bench.py -- Kalman log-likelihood evals/s, CARMA(5,3), n=270 (BASELINE.json metric).

A "step" is one pass of the hot path over one batch: B=1024 batched CARMA_Base::LogDensity
evaluations (BASELINE configs[1]) on the README synthetic series, parameter vectors already
resident in HBM, one kernel launch through the C ABI (carma_logdensity_batch_dev).

    python bench.py --gpus 1 --steps 2000 --warmup 50
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU.  The batch is sharded by rank with no data-path collective (independent
evaluations -> weak scaling: every rank runs its own 1024-eval batches); RCCL is used only for
the barrier and the max-over-ranks of the elapsed time.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402  (import torch BEFORE the HIP library so both share one HIP runtime)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
FP64_VALU_PEAK_TFLOPS = 78.6   # public MI355X FP64 vector peak = 256 CUs x 4 SIMDs x 16 FMA lanes/clk x 2 flop x 2.4 GHz
MAX_CLOCK_GHZ = 2.4            # MI355X_MICROARCH.md: max clock; a wave64 VALU instruction holds its SIMD's issue port 4 cycles


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--batch", type=int, default=1024, help="evals per step (BASELINE config 2: 1024)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-mcmc", action="store_true", help="skip the secondary MCMC iterations/s leg")
    ap.add_argument("--mcmc-iters", type=int, default=2000)
    ap.add_argument("--no-mcmc-large", action="store_true", help="skip the large-ensemble sampler leg (16 x 4096 ladders)")
    ap.add_argument("--graph", action="store_true", help="replay the steps as a hipGraph instead of launching each directly "
                    "(measured: 34.43 vs 34.66 us per step -- the gap between dependent kernels is not the host's)")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the two-batches-in-flight leg")
    ap.add_argument("--no-steady", action="store_true", help="skip the steady-state leg (the headline step 1000 times)")
    ap.add_argument("--no-throughput", action="store_true", help="skip the throughput-regime leg (B = 65536)")
    ap.add_argument("--no-ladder", action="store_true", help="skip the ladder-sharded leg (BASELINE configs[3] shape)")
    ap.add_argument("--ladder-timeout", type=float, default=180.0,
                    help="N > 1: seconds after which the run ends without the ladder-sharded leg (it is the only leg with an "
                         "exchange between the ranks)")
    ap.add_argument("--ladder-iters", type=int, default=150)
    ap.add_argument("--no-api", action="store_true", help="skip the legs through the drop-in Python API (the reference's own workloads: "
                    "the notebook's run_mcmc(20000) on OGLE, the README's run_mcmc(50000), choose_order(pmax=7))")
    ap.add_argument("--api-scale", type=float, default=1.0, help="fraction of the API legs' iteration counts (tests use 0.02)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node == --gpus"
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    # Test hook for a one-GPU box: CARMA_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and moves the barrier /
    # max-time reduction to gloo (RCCL refuses two ranks on one device).  The driver never sets it.
    share = os.environ.get("CARMA_BENCH_SHARE_GPU") == "1"
    if share:
        # k_pt_row spreads a ladder over several workgroups that meet at a rendezvous and needs them all resident:
        # true on a GPU of its own, not when two processes time-share one (the launch would abort with an error)
        os.environ["CARMA_PT_KERNEL"] = "ladder"
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    import carma_pack_amd as cpa
    from carma_pack_amd.synth import theta_batch

    # who is here: every rank's device, gathered once (a driver-run SCALE record shows at a glance that N ranks on N
    # different devices took part, and which RCCL carried the barrier)
    me = {"rank": rank, "local_rank": local_rank, "device_ordinal": dev_index, "device_name": torch.cuda.get_device_name(dev_index),
          "pci_bus_id": _pci_bus_id(dev_index)}
    ranks_info = [me]
    if dist is not None:
        ranks_info = [None] * world
        dist.all_gather_object(ranks_info, me)
    try:
        rccl_version = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:                                          # noqa: BLE001 (a build without it)
        rccl_version = None

    g = np.load(os.path.join(ROOT, "tests", "golden", "carma53_readme.npz"))
    t, y, yerr = g["t"], g["y"], g["yerr"]
    p, q = 5, 3
    n, d, B = t.size, 3 + p + q, args.batch
    max_stdev = 10.0 * np.sqrt(np.mean(y * y) - np.mean(y) ** 2)      # src/carmcmc.cpp:85-89
    ctx = cpa.Context(t, y, yerr, p, q, max_stdev=max_stdev, device=dev_index)

    # synthetic parameter vectors: posterior-like + prior-like, 8 distinct
    # batches per rank so consecutive steps never see the same input.
    NPOOL = 8
    rng = np.random.default_rng(2 + 1000 * rank)
    pool_h = [theta_batch(rng, B, p, q, t, y, theta_center=g["theta"][0]) for _ in range(NPOOL)]
    pool = [torch.from_numpy(a).to(dev) for a in pool_h]
    out = torch.empty(B, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream()
    sh = stream.cuda_stream

    def step(i):
        ctx.logdensity_dev(pool[i % NPOOL].data_ptr(), B, out.data_ptr(), ignore_prior=False, stream=sh)

    def barrier():
        if dist is not None:
            dist.barrier()

    # ORDER OF THE LEGS.  The secondary legs (sampler, two batches in flight, throughput regime) run FIRST, the headline
    # region -- W untimed warm-up steps, then exactly K timed steps -- after them: a run of `--steps 20 --warmup 5` is 25
    # launches of 30 us, far too short for a GPU that has just been woken up to reach the clocks it holds under this
    # load (round 2: 33.1 us per step in such a run against 29.7 us in a 2000-step run).  The legs are the same work a
    # user of the library would have had the device do; the headline region itself is unchanged.
    # The sampler leg goes LAST, right in front of the warm-up: it is the same kind of load as the headline kernel (one
    # four-wave workgroup per CU), 60 ms of it, whereas the throughput leg fills every SIMD with FP64 work and leaves the
    # clocks where that load's power allows -- a kernel trace of a 20-step run (tools/trace_short_bench.sh) showed the
    # headline launches at 31.0 us right behind it, drifting down by 0.01 us per launch.
    # ---- the same steps with TWO batches in flight (two streams, alternating): what a caller with independent batches
    # should do -- the second launch fills the half of every CU's issue slots that one four-wave workgroup leaves idle.
    # Extra key; the headline `value` above is one batch at a time on one stream.
    pipelined = None
    if not args.no_pipelined:
        s2 = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
        outs = [out, torch.empty(B, dtype=torch.float64, device=dev)]
        torch.cuda.synchronize()
        for i in range(8):
            ctx.logdensity_dev(pool[i % NPOOL].data_ptr(), B, outs[i & 1].data_ptr(), ignore_prior=False, stream=s2[i & 1].cuda_stream)
        torch.cuda.synchronize()
        barrier()
        # (a secondary leg with its own launch count: 20 launches would not amortise the closing synchronisation of two streams)
        PSTEPS = max(args.steps, 200)
        tp0 = time.perf_counter()
        for i in range(PSTEPS):
            ctx.logdensity_dev(pool[i % NPOOL].data_ptr(), B, outs[i & 1].data_ptr(), ignore_prior=False, stream=s2[i & 1].cuda_stream)
        torch.cuda.synchronize()
        tp = time.perf_counter() - tp0
        barrier()
        if dist is not None:
            tt = torch.tensor([tp], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            tp = float(tt.item())
        pipelined = {"metric": "Kalman log-lik evals/sec, the same %d-evaluation batches, two in flight (two streams)" % B,
                     "evals_per_s": world * B * PSTEPS / tp, "ms_per_step": 1e3 * tp / PSTEPS, "streams": 2,
                     "steps": PSTEPS}

    # ---- throughput regime: the same kernel family with the chip full (B = 65536 per launch) ----------------
    tput = None
    if not args.no_throughput:
        BT = 65536
        big = torch.from_numpy(np.tile(pool_h[0], (BT // B + 1, 1))[:BT].copy()).to(dev)
        outb = torch.empty(BT, dtype=torch.float64, device=dev)
        # (40 untimed launches, 9 ms: with every SIMD full of FP64 work the clocks settle over milliseconds -- behind a 20-step
        # run the first launches of this leg took 240 us where they take 216 behind a 200-step run, profiles/r05/bench_*_v4.json)
        for _ in range(40):
            ctx.logdensity_dev(big.data_ptr(), BT, outb.data_ptr(), ignore_prior=False, stream=sh)
        torch.cuda.synchronize()
        barrier()
        NT = 20
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        tq0 = time.perf_counter()
        e0.record(stream)
        for _ in range(NT):
            ctx.logdensity_dev(big.data_ptr(), BT, outb.data_ptr(), ignore_prior=False, stream=sh)
        e1.record(stream)
        torch.cuda.synchronize()
        tq = time.perf_counter() - tq0
        barrier()
        if dist is not None:
            tt = torch.tensor([tq], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            tq = float(tt.item())
        k_us = 1e3 * e0.elapsed_time(e1) / NT
        tput = {
            "metric": "Kalman log-lik evals/sec with the chip full (same series, %d evaluations per launch)" % BT,
            "evals_per_s": world * BT * NT / tq, "batch_per_gpu": BT, "launches": NT, "kernel": ctx.kernel_name(BT),
            "kernel_avg_us": k_us,
        }

    # ---- the same with 2^20 evaluations in ONE launch: where the one-evaluation-per-lane kernel reaches its best share of the
    # chip's FP64 issue slots (four waves per SIMD) -- five launches, so the figure is driver-run
    tput1m = None
    if not args.no_throughput:
        BM, NM = 1 << 20, 5
        bigm = torch.from_numpy(np.tile(pool_h[1], (BM // B + 1, 1))[:BM].copy()).to(dev)
        outm = torch.empty(BM, dtype=torch.float64, device=dev)
        ctx.logdensity_dev(bigm.data_ptr(), BM, outm.data_ptr(), ignore_prior=False, stream=sh)
        torch.cuda.synchronize()
        barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        tq0 = time.perf_counter()
        e0.record(stream)
        for _ in range(NM):
            ctx.logdensity_dev(bigm.data_ptr(), BM, outm.data_ptr(), ignore_prior=False, stream=sh)
        e1.record(stream)
        torch.cuda.synchronize()
        tq = time.perf_counter() - tq0
        barrier()
        if dist is not None:
            tt = torch.tensor([tq], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            tq = float(tt.item())
        tput1m = {
            "metric": "Kalman log-lik evals/sec, %d evaluations per launch" % BM,
            "evals_per_s": world * BM * NM / tq, "batch_per_gpu": BM, "launches": NM, "kernel": ctx.kernel_name(BM),
            "kernel_avg_us": 1e3 * e0.elapsed_time(e1) / NM, "finite": int(torch.isfinite(outm).sum().item()),
        }
        del bigm, outm

    # ---- the sampler with a LARGE ensemble (before the configs[2]-shape leg, which stays last: see ORDER OF THE LEGS): 16 temperatures x 4096 ladders = 65 536 chains per GPU, one chain per lane,
    # an iteration = propose kernel + the batched log-density launch + finish kernel (carma_pt_lane.hip)
    mcmc_large = None
    if not args.no_mcmc and not args.no_mcmc_large:
        T_, R_, IT_ = 16, 4096, 60
        big_ctx = cpa.Context(t, y, yerr, p, q, max_stdev=max_stdev, device=dev_index)
        big_ctx.pt_create(T_, R_, adapt_iters=10 ** 9, seed=13 + rank)
        big_ctx.pt_shard(T_, 0, rank * R_)
        big_ctx.pt_start(None)
        big_ctx.pt_iterate(10)
        barrier()
        tm0 = time.perf_counter()
        big_ctx.pt_iterate(IT_)
        tm = time.perf_counter() - tm0
        barrier()
        if dist is not None:
            tt = torch.tensor([tm], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            tm = float(tt.item())
        acc_, swp_ = big_ctx.pt_stats()
        mcmc_large = {
            "metric": "MCMC iterations/s, large ensemble (each iteration advances every chain once)",
            "iters_per_s": IT_ / tm, "chain_evals_per_s": world * T_ * R_ * IT_ / tm, "us_per_iteration": 1e6 * tm / IT_,
            "temperatures": T_, "replicas_per_gpu": R_, "chains_per_gpu": T_ * R_, "iters": IT_, "sampler": big_ctx.pt_kernel(),
            "log_density_kernel": big_ctx.kernel_name(T_ * R_),
            "accept_rate": float(acc_.mean()), "swap_rate": float(swp_[:, 1:].mean()),
        }
        del big_ctx

    # ---- secondary metric: MCMC iterations/s, BASELINE configs[2] shape ------------------------
    # 16 temperatures x 64 independent ladders ("walkers") per GPU, persistent PT kernel; one
    # iteration = every chain does one RAM step (one Kalman eval) + one exchange sweep.
    mcmc = None
    if not args.no_mcmc:
        T_, R_ = 16, 64
        ctx.pt_create(T_, R_, adapt_iters=10 ** 9, seed=11 + rank)
        ctx.pt_shard(T_, 0, rank * R_)
        ctx.pt_start(None)
        ctx.pt_iterate(100)
        barrier()
        tm0 = time.perf_counter()
        ctx.pt_iterate(args.mcmc_iters)
        tm = time.perf_counter() - tm0
        barrier()
        if dist is not None:
            tt = torch.tensor([tm], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            tm = float(tt.item())
        acc_, swp_ = ctx.pt_stats()
        mcmc = {
            "metric": "MCMC iterations/s (each iteration advances every chain once)",
            "iters_per_s": args.mcmc_iters / tm,
            "chain_evals_per_s": world * T_ * R_ * args.mcmc_iters / tm,
            "temperatures": T_, "replicas_per_gpu": R_, "iters": args.mcmc_iters,
            "accept_rate": float(acc_.mean()), "swap_rate": float(swp_[:, 1:].mean()),
            "config": "configs[2] shape: CARMA(5,3), n=270, 16 temperatures x 64 walkers per GPU, RAM adapting",
        }

    # ---- the headline launch in steady state: the same step() 1000 times back to back, as a figure of its own.  (The timed region
    # below is K launches behind W warm-up launches as the contract says; with the driver's K = 20 that is 0.6 ms of a kernel, and the
    # figure it gives moves by several per cent with whatever the device did in the millisecond before -- rounds 2-4, NOTEBOOK.md.
    # This leg is the same work measured over 30 ms.)
    steady = None
    if not args.no_steady:
        NS = 1000
        for i in range(8):
            step(i)
        torch.cuda.synchronize()
        barrier()
        ts0 = time.perf_counter()
        for i in range(NS):
            step(i)
        torch.cuda.synchronize()
        ts = time.perf_counter() - ts0
        barrier()
        if dist is not None:
            tt = torch.tensor([ts], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            ts = float(tt.item())
        steady = {"metric": "Kalman log-lik evals/sec, the headline step %d times back to back" % NS, "evals_per_s": world * B * NS / ts,
                  "ms_per_step": 1e3 * ts / NS, "launches": NS}

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()

    # --graph: the K steps are K launches of the same kernel on one stream; they can be captured ONCE as a hipGraph of
    # GRAPH_N consecutive steps (the C ABI only enqueues, so stream capture records its launches) and replayed.  Empty
    # kernels start 1.6 us after their predecessor as graph nodes instead of 2.7 us (tools/ubench/launch_gap.hip), but on
    # the real kernel the step time moves by < 1 %: the host is not what separates dependent launches.  Off by default.
    GRAPH_N = 25 * NPOOL
    graph = None
    if args.graph and args.steps >= GRAPH_N:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            cap = torch.cuda.current_stream().cuda_stream
            for i in range(GRAPH_N):
                ctx.logdensity_dev(pool[i % NPOOL].data_ptr(), B, out.data_ptr(), ignore_prior=False, stream=cap)
        graph.replay()                      # untimed: instantiation / upload of the graph, not a step of the K
        torch.cuda.synchronize()

    # ---- timed region: EXACTLY K steps --------------------------------------------------------
    # (N > 1: every rank reads its clock when ITS K steps are through -- between the closing synchronize and the closing barrier -- and the
    # MAX over ranks is the job's time: all ranks left the opening barrier together, so that is when the last one finished.  The closing
    # barrier itself -- an RCCL all-reduce and its host wake-up, ~0.1 ms -- is not a step of the path: inside a 20-step region of 0.55 ms it
    # would read as a fifth of the time.  The legs above are timed the same way.  N = 1: the barrier is a no-op.)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record(stream)
    nrep = args.steps // GRAPH_N if graph is not None else 0
    for _ in range(nrep):
        graph.replay()
    for i in range(nrep * GRAPH_N, args.steps):
        step(i)
    ev1.record(stream)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    dev_ms = ev0.elapsed_time(ev1)
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # sanity: the last step's output is the real thing
    last = out.cpu().numpy()
    n_finite = int(np.isfinite(last).sum())

    # ---- per-launch kernel duration (HIP events on the launch stream) ------------------------
    M = max(1, min(args.steps, 256))
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(M)]
    for i, (a, b) in enumerate(evs):
        a.record(stream)
        step(i)
        b.record(stream)
    torch.cuda.synchronize()
    kdur_ms = np.array([a.elapsed_time(b) for a, b in evs])
    # Average launch duration of the dominant kernel: HIP events around the K back-to-back launches of the timed region
    # (same stream), divided by K -- this is what rocprofv3's per-kernel average agrees with (profiles/).  Bracketing
    # every launch with its own pair of events (kdur_ms) adds the markers' own time (~2 us) to each figure; kept as
    # "kernel_bracketed_*" for reference.
    kernel_ms = dev_ms / args.steps

    res = None
    if rank == 0:
        value = world * B * args.steps / elapsed
        bytes_per_eval = 24 * n + 8 * d + 8
        flops_per_eval = (n - 1) * (42 * p * p + 22 * p + 9)
        achieved_gbs = bytes_per_eval * B / (kernel_ms * 1e-3) / 1e9
        # HBM bytes per launch: the newest committed rocprofv3 PMC summary (profiles/r*/pmc_*.json, written by
        # tools/profile_round.sh + tools/summarize_prof.py from separate --pmc passes of this same command) whose
        # dispatch record names the kernel this run actually launched; none -> null, never a stale figure.
        # (gfx950 FETCH_SIZE can under-count wide coalesced reads by 2x, so 2*FETCH+WRITE is the upper bound.)
        kernel_name = ctx.kernel_name(B)
        ids = cpa._lib.build_ids()
        traffic_gbs_bytes, traffic_src, pmc_extra = None, None, None
        # (grid in threads: the two-sided kernel takes two evaluations per 256-thread workgroup, the other wave pipelines four)
        wgs = (B + 1) // 2 if kernel_name.startswith("k_logdens_carma_w2<") else (B + 3) // 4
        pj, pmc_path = committed_pmc(kernel_name, wgs * 256, ids)                   # same kernel, same launch shape, same build
        if pj is not None and "FETCH_SIZE" in pj and "WRITE_SIZE" in pj:
            traffic_gbs_bytes = (2.0 * pj["FETCH_SIZE"]["mean"] + pj["WRITE_SIZE"]["mean"]) * 1024.0
            traffic_src = "%s (rocprofv3 --pmc of this command on this build, not live): upper bound 2*FETCH_SIZE+WRITE_SIZE bytes per launch" % pmc_path
        elif pj is None:
            traffic_src = pmc_path                               # why there is none (no record / another build)
        pmc_extra = pmc_digest(pj, pmc_path, 1e3 * kernel_ms)
        if mcmc is not None:
            # per ITERATION: the iteration's wall time is what the counters are set against
            mcmc["pmc_per_iteration"] = pmc_digest(*committed_pmc("k_pt_row<%d," % p, None, ids), 1e6 / mcmc["iters_per_s"])
        if tput is not None:
            tput["pmc_per_launch"] = pmc_digest(*committed_pmc(tput["kernel"], tput["batch_per_gpu"], ids), tput["kernel_avg_us"])
            ceilings(tput, tput["pmc_per_launch"])
        if tput1m is not None:
            # counters scale with the number of evaluations (same kernel, same per-wave instruction stream): the 65 536-evaluation
            # record, scaled by the ratio of the batch sizes, against THIS leg's launch time
            pj1, path1 = committed_pmc(tput1m["kernel"], tput1m["batch_per_gpu"], ids)      # counters of the 2^20-evaluation launches themselves
            if pj1 is not None:
                tput1m["pmc_per_launch"] = pmc_digest(pj1, path1, tput1m["kernel_avg_us"])
                tput1m["ceilings_source"] = "measured at this launch size"
            else:
                tput1m["pmc_per_launch"] = pmc_digest(*committed_pmc(tput1m["kernel"], 65536, ids), tput1m["kernel_avg_us"],
                                                      scale=tput1m["batch_per_gpu"] / 65536.0)
                tput1m["ceilings_source"] = "scaled from the 65 536-evaluation record"
            ceilings(tput1m, tput1m["pmc_per_launch"])
        if mcmc is not None:
            ceilings(mcmc, mcmc["pmc_per_iteration"])
        res = {
            "metric": "Kalman log-lik evals/sec, CARMA(5,3) n=270",
            "value": value,
            "unit": "evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "configs[1]: CARMA(5,3), n=270 README synthetic series, %d batched "
                            "log-density evals per step (one launch), thetas resident in HBM" % B,
                "submission": ("hipGraph of %d consecutive steps, replayed %d times + %d direct launches" % (
                    GRAPH_N, nrep, args.steps - nrep * GRAPH_N)) if graph is not None else "one direct launch per step",
                "p": p, "q": q, "n": int(n), "batch_per_gpu": B, "parallelism": "batch sharded by rank, no collective",
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved_gbs,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved_gbs / HBM_PEAK_GBS,
                "traffic": traffic_gbs_bytes,
                "traffic_source": traffic_src,
                "kernel": kernel_name,
                "kernel_avg_us": 1e3 * kernel_ms,
                "kernel_bracketed_avg_us": 1e3 * float(np.mean(kdur_ms)),
                "kernel_bracketed_min_us": 1e3 * float(kdur_ms.min()),
                "algorithmic_bytes_per_launch": bytes_per_eval * B,
                "binding_resource": "fp64_valu_issue",
                "note": "HBM roofline reported as north_star asks; the kernel is bound by the FP64 VALU issue rate of one "
                        "wave's dependent instruction stream (sequential n-step recursion): pmc_per_launch.valu_issue_frac is the "
                        "measured share of the chip's VALU issue slots, pmc_per_launch.fp64 the executed flops",
                "pmc_per_launch": pmc_extra,
                "algorithmic_flops_per_eval_reference_complex_count": flops_per_eval,
            },
            "build": ids,
            "backend": ("gloo (test hook: ranks share one GPU)" if share else "nccl (RCCL)") if world > 1 else "none (one rank)",
            "rccl_version": rccl_version,
            "ranks": ranks_info,
            "finite_in_last_batch": n_finite,
            "timing": "K steps between torch.cuda.synchronize() calls; every rank reads its clock BEFORE the closing barrier, MAX over ranks "
                      "(the barrier's own latency is not a step: changed in round 5, see NOTEBOOK.md)",
        }
        if mcmc is not None:
            res["mcmc"] = mcmc
        ceilings(res["roofline"], pmc_extra)
        if mcmc_large is not None:
            res["mcmc_large"] = mcmc_large
        if steady is not None:
            res["steady_state"] = steady
        if pipelined is not None:
            res["pipelined"] = pipelined
        if tput is not None:
            res["throughput"] = tput
        if tput1m is not None:
            res["throughput_1m"] = tput1m

    # ---- ONE ladder sharded across the ranks (BASELINE configs[3]: CARMA(7,6), n = 10^4, 8 temperatures) -------
    # Last, and at N > 1 under a watchdog: this leg is the only one with a data-path exchange (RCCL send/recv between the
    # ranks), and the headline line must not depend on it -- neither on an exception nor on a collective that never
    # returns.  If the leg has not finished in time every rank leaves; rank 0 prints the line without it first.
    ladder = None
    if not args.no_ladder and 8 % world == 0:
        LADDER_TEXT = "MCMC iterations/s of ONE temperature ladder sharded across the ranks"
        watchdog = None
        printed = None
        if world > 1:
            import threading
            printed = threading.Lock()                     # whoever takes it prints the line; nobody prints twice

            def give_up():
                # The exchange hangs: rank 0 prints the line WITH the failure in it ("ladder_leg_hung": true and the leg's
                # error; the headline measurement of this N was complete before the leg started), says so on stderr, and
                # EVERY rank leaves with a non-zero status -- a wedged RCCL exchange is a failed run to the launcher; a
                # harness that wants the partial numbers reads the line on rc != 0.  No retry in here: a process that has
                # touched the GPU is never re-executed, a fresh process is the only retry.
                if rank == 0 and printed.acquire(blocking=False):
                    out = dict(res)
                    out["ladder_leg_hung"] = True
                    out["ladder_sharded"] = {"metric": LADDER_TEXT, "error": "no result after %d s" % args.ladder_timeout}
                    sys.stderr.write("bench.py: the ladder-sharded leg did not return within %d s; leaving with status 3\n" % args.ladder_timeout)
                    print(json.dumps(out), flush=True)
                    sys.stdout.flush()
                elif rank != 0:
                    sys.stderr.write("bench.py: rank %d: the ladder-sharded leg did not return within %d s; leaving with status 3\n" % (rank, args.ladder_timeout))
                sys.stderr.flush()
                os._exit(3)

            watchdog = threading.Timer(args.ladder_timeout, give_up)
            watchdog.daemon = True
            watchdog.start()
        try:
            ladder = ladder_sharded_leg(cpa, dist, world, share, dev, dev_index, barrier, args.ladder_iters)
        except Exception as ex:
            ladder = {"metric": LADDER_TEXT, "error": repr(ex)}
        if watchdog is not None:
            watchdog.cancel()
            if rank == 0 and not printed.acquire(blocking=False):
                time.sleep(60)                             # the watchdog fired while the leg was returning: it is leaving

    if rank == 0:
        if ladder is not None:
            res["ladder_sharded"] = ladder
            # first contact of the sharded ladder with RCCL, at top level: did N ranks exchange, and was it right
            fc = ladder.get("first_contact_check") or {}
            res["rccl_first_contact"] = {
                "ranks_in_exchange": ladder.get("rccl_ranks"),
                "verdict": ("error: " + ladder["error"]) if "error" in ladder else
                           ("one rank: no exchange" if world == 1 else
                            ("ok" if fc.get("equals_one_gpu_run") in (True, None) and all(v == 1 for v in (fc.get("boundary_checksums_agree") or []) if v is not None)
                             and fc.get("equals_one_gpu_run") is not False else "MISMATCH")),
                "equals_one_gpu_run": fc.get("equals_one_gpu_run"), "kernels_compared": fc.get("kernels_compared"),
                "boundary_checksums_agree": fc.get("boundary_checksums_agree"), "transport": ladder.get("transport"),
            }
        if world == 1 and not args.no_cpu:
            res["cpu_baseline"] = cpu_baseline(t, y, yerr, p, q, max_stdev, pool_h[0], args.cpu_seconds)
        if world == 1 and not args.no_api:
            try:
                res.update(api_legs(args.api_scale, None if args.no_cpu else res.get("cpu_baseline")))
            except Exception as ex:                                   # noqa: BLE001 (the headline line does not depend on these legs)
                res["api_legs_error"] = repr(ex)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def api_legs(scale, cpu):
    """The reference's OWN workloads through the drop-in Python API (carmcmc.CarmaModel, not the C ABI):
      quickstart_ogle   CarmaModel(t, y, yerr, p=6, q=0).run_mcmc(20000) on OGLE-LMC-LPV-00007 -- the notebook's "about 10-15 minutes"
                        (examples/carma_pack_guide.ipynb:302-327): 30 000 iterations of ONE ladder of 10 temperatures
      readme_run        CarmaModel(p=5, q=3).run_mcmc(50000) on the README series (README.md:65-71): 75 000 iterations x 10 temperatures
      choose_order      choose_order(pmax=7, ntrials=100) on OGLE (28 orders, carma_pack.py:131-192)
    each with the wall time split into: the sampler call (run_mcmc_carma: start values, chains on the device, samples to the host),
    the post-hoc log-likelihood batch (carma_pack.py:305-315), the sigma_noise batch, and the rest of CarmaSample's construction;
    beside it the cpu_baseline port's single-thread estimate for the same number of Kalman evaluations."""
    import carmcmc as cm
    from carma_pack_amd import _carmcmc as lib
    from carma_pack_amd import carma_pack as cp
    og = np.loadtxt(os.path.join(ROOT, "tests", "golden", "ogle_lmc_lpv_00007.dat"))
    to, yo, eo = og[:, 0] - og[:, 0].min(), og[:, 1], og[:, 2]
    g = np.load(os.path.join(ROOT, "tests", "golden", "carma53_readme.npz"))
    cpu1 = (cpu or {}).get("single_thread_value")

    def timed_run(t, y, e, p, q, nsamples):
        acc = {"sampler": 0.0, "loglik_batch": 0.0, "sigma_noise": 0.0}
        orig_run, orig_sig = lib.run_mcmc_carma, lib.sigma_noise_batch

        def run(*a, **k):
            t0 = time.perf_counter()
            obj = orig_run(*a, **k)
            acc["sampler"] += time.perf_counter() - t0
            ob = obj.getLogDensityBatch

            def batch(*aa, **kk):
                t1 = time.perf_counter()
                r = ob(*aa, **kk)
                acc["loglik_batch"] += time.perf_counter() - t1
                return r
            obj.getLogDensityBatch = batch
            return obj

        def sig(*a, **k):
            t0 = time.perf_counter()
            r = orig_sig(*a, **k)
            acc["sigma_noise"] += time.perf_counter() - t0
            return r
        lib.run_mcmc_carma, lib.sigma_noise_batch = run, sig
        try:
            t0 = time.perf_counter()
            model = cm.CarmaModel(t, y, e, p=p, q=q)
            sample = model.run_mcmc(nsamples, seed=20260101)
            wall = time.perf_counter() - t0
        finally:
            lib.run_mcmc_carma, lib.sigma_noise_batch = orig_run, orig_sig
        ntemp = max(10, p + q)
        iters = nsamples + nsamples // 2
        evals = iters * ntemp + nsamples
        lp = sample.get_samples("logpost")
        out = {"call": "CarmaModel(t, y, yerr, p=%d, q=%d).run_mcmc(%d)" % (p, q, nsamples), "n": int(len(t)), "wall_s": wall,
               "iterations": iters, "temperatures": ntemp, "iterations_per_s": iters / acc["sampler"], "kalman_evals": evals,
               "split_s": {"sampler_call": acc["sampler"], "post_hoc_loglik_batch": acc["loglik_batch"], "sigma_noise_batch": acc["sigma_noise"],
                           "carma_sample_rest": wall - sum(acc.values())},
               "samples": int(lp.size), "logpost_finite": bool(np.all(np.isfinite(lp)))}
        if cpu1:
            # the CPU port's single-thread rate is for CARMA(5,3), n = 270: scaled by the step count n p^2 of this call
            rate = cpu1 * (270.0 * 25.0) / (len(t) * float(p * p))
            out["cpu_port_single_thread_estimate_s"] = evals / rate
        return out

    res = {}
    ns1, ns2 = max(200, int(20000 * scale)), max(200, int(50000 * scale))
    res["quickstart_ogle"] = timed_run(to, yo, eo, 6, 0, ns1)
    res["quickstart_ogle"]["reference_statement"] = "about 10-15 minutes (examples/carma_pack_guide.ipynb:302-327; hardware not stated)"
    res["readme_run"] = timed_run(g["t"], g["y"], g["yerr"], 5, 3, ns2)
    t0 = time.perf_counter()
    model = cm.CarmaModel(to, yo, eo)
    ntr = max(4, int(100 * min(1.0, scale * 5)))
    best, pqlist, aicc = model.choose_order(7, ntrials=ntr, seed=5)
    res["choose_order"] = {"call": "CarmaModel(t, y, yerr).choose_order(7, ntrials=%d) on OGLE-LMC-LPV-00007" % ntr, "wall_s": time.perf_counter() - t0,
                           "orders": len(pqlist), "chosen": [int(model.p), int(model.q)], "aicc_min": float(np.min(aicc))}
    if scale != 1.0:
        for k in res:
            res[k]["scale"] = scale
    return res


def _pci_bus_id(dev_index):
    try:
        pr = torch.cuda.get_device_properties(dev_index)
        return "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
    except Exception:                                          # noqa: BLE001
        return None


def committed_pmc(kernel_substr, grid=None, ids=None):
    """Counters of `kernel_substr` from the newest committed rocprofv3 PMC summary (profiles/r*/pmc_*.json, written by
    tools/profile_round.sh + tools/summarize_prof.py from separate --pmc passes of this command) whose dispatch record
    names that kernel (and launch grid) AND whose build stamp is the running library's: the same binary (build_id) or a
    binary built from the same sources (source_id).  (None, reason) otherwise -- never a stale or foreign figure."""
    import glob
    import re
    # newest = highest (round, version) in the path (profiles/r03/pmc_v2.json); file times mean nothing after a checkout
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_*.json")),
                   key=lambda f: [int(x) for x in re.findall(r"\d+", os.path.relpath(f, ROOT))], reverse=True)
    seen_other_build = None
    for pmc in cands:
        try:
            pj = json.load(open(pmc))
        except ValueError:
            continue
        disp = pj.get("_dispatch") or {}
        # (rocprofv3 prints defaulted template arguments too: "k_logdens_carma<5, 8, 4, false>" for "k_logdens_carma<5,8,4>")
        if kernel_substr.replace(" ", "").rstrip(">") not in disp.get("Kernel_Name", "").replace(" ", ""):
            continue
        if grid is not None and str(disp.get("Grid_Size", "")) != str(grid):
            continue
        stamp = pj.get("_ids") or {}
        if ids is not None and not ((stamp.get("build_id") and stamp.get("build_id") == ids.get("build_id")) or
                                    (stamp.get("source_id") and stamp.get("source_id") == ids.get("source_id"))):
            seen_other_build = seen_other_build or os.path.relpath(pmc, ROOT)
            continue
        return pj, os.path.relpath(pmc, ROOT)
    if seen_other_build:
        return None, "none: the newest record of this kernel (%s) was measured on another build of the library" % seen_other_build
    return None, "none: no committed counter record of this kernel"


def ceilings(block, digest):
    """The ceilings that actually bind this path, as SCALARS next to the block's other numbers (a harness that keeps scalars
    and drops nested objects still sees them): share of the chip's VALU issue slots, executed FP64 flops against the vector
    peak.  None when no counter record of the running build exists."""
    d = digest if isinstance(digest, dict) else {}
    fp = d.get("fp64") or {}
    block["valu_issue_frac"] = d.get("valu_issue_frac")
    block["fp64_frac"] = fp.get("frac")
    block["fp64_tflops"] = fp.get("achieved_tflops")
    block["fp64_peak_tflops"] = FP64_VALU_PEAK_TFLOPS


def pmc_digest(pj, src, wall_us=None, scale=1.0):
    """The counters of one record, per launch (or per sampler iteration), and what follows from them and from the wall
    time `wall_us` measured in THIS run:
      valu_issue_frac = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x wall x 2.4 GHz)   -- share of the chip's VALU issue slots
                        (every wave64 VALU instruction holds its SIMD's issue port for 4 cycles; FP64 DPP and
                        transcendentals longer, so this is a lower bound of the port's busy share and can never exceed 1)
      fp64: executed flops = (2 FMA + ADD + MUL instructions) x 64 lanes, as the hardware counts instructions (idle lanes
            of a partly filled wave included) -- not the reference's complex-arithmetic count."""
    if pj is None:
        return {"source": src} if src else None
    out = {k: pj[k]["mean"] * scale for k in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES",
                                      "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU_FMA_F64",
                                      "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_TRANS_F64", "GRBM_GUI_ACTIVE")
           if k in pj}
    out["per"] = pj.get("_per", "launch")
    if scale != 1.0:
        out["scaled_by"] = scale
    out["kernel"] = (pj.get("_dispatch") or {}).get("Kernel_Name", "").split("(")[0].replace("void carma::", "")
    out["vgprs"] = (pj.get("_dispatch") or {}).get("VGPR_Count")
    out["source"] = "%s (rocprofv3 --pmc of this command, not live; FETCH/WRITE in KiB)" % src
    out["measured_on"] = pj.get("_ids")
    if wall_us and "SQ_INSTS_VALU" in out:
        simds = 256 * 4
        out["valu_issue_frac"] = out["SQ_INSTS_VALU"] * 4.0 / (simds * wall_us * 1e-6 * MAX_CLOCK_GHZ * 1e9)
        out["valu_issue_frac_basis"] = "SQ_INSTS_VALU x 4 cycles / (%d SIMDs x %.2f us x %.1f GHz max clock)" % (simds, wall_us, MAX_CLOCK_GHZ)
    if wall_us and all(k in out for k in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64")):
        fl = (2.0 * out["SQ_INSTS_VALU_FMA_F64"] + out["SQ_INSTS_VALU_ADD_F64"] + out["SQ_INSTS_VALU_MUL_F64"]) * 64.0
        tf = fl / (wall_us * 1e-6) / 1e12
        out["fp64"] = {"executed_flops": fl, "achieved_tflops": tf, "peak_tflops": FP64_VALU_PEAK_TFLOPS, "frac": tf / FP64_VALU_PEAK_TFLOPS,
                       "basis": "(2 x SQ_INSTS_VALU_FMA_F64 + _ADD_F64 + _MUL_F64) x 64 lanes / wall time of this run"}
    return out


def ladder_sharded_leg(cpa, dist, world, share, dev, dev_index, barrier, iters):
    """N = 1: the whole ladder on one GPU.  N > 1: contiguous temperature blocks per rank, boundary chains exchanged with
    RCCL send/recv on the sampler's stream (carma_pt_iterate_sharded).  Iterations/s of the whole ladder (strong scaling:
    the ladder is the same at every N).  At N > 1 the first ten iterations are also CHECKED: the blocks' chain states,
    gathered on rank 0, must equal those of the same ladder run on rank 0's GPU alone (the sharded ladder walks the
    unsharded trajectory bit for bit), and every boundary's decision checksums must have agreed."""
    from carma_pack_amd import _lib, parallel as par
    from carma_pack_amd.synth import config4_series
    t4, y4, e4, _ = config4_series(10000, seed=4)
    TG, R4, SEED4, WARM = 8, 128, 17, 10
    ctx4 = cpa.Context(t4, y4, e4, 7, 6, device=dev_index)
    sh4 = par.LadderShard(ctx4, TG, R4, adapt_iters=10 ** 9, seed=SEED4, dist=dist if world > 1 else None, device="cpu")
    comm = None
    # (ranks sharing one GPU -- the test hook: RCCL refuses, the torch.distributed stand-in runs; unless CARMA_RCCL_LIB names
    # another transport for the library's native path, tests/shm_transport)
    if world > 1 and (not share or os.environ.get("CARMA_RCCL_LIB")):
        comm = _lib.Comm.from_torch(dist, device=dev_index)
        sh4.attach_comm(comm)
    sh4.start()
    run = sh4.iterate if world > 1 else ctx4.pt_iterate
    run(WARM)
    check = None
    if world > 1:
        th, lp = ctx4.pt_get_chains()
        parts = [None] * world
        dist.all_gather_object(parts, (th, lp, ctx4.pt_boundary_check() if comm is not None else None, ctx4.pt_kernel()))
        if dist.get_rank() == 0:
            one = cpa.Context(t4, y4, e4, 7, 6, device=dev_index)
            one.pt_create(TG, R4, 10 ** 9, seed=SEED4, temperatures=par.ladder_temperatures(TG))
            one.pt_shard(TG, 0, 0)
            one.pt_start(None)
            one.pt_iterate(WARM)
            uth, ulp = one.pt_get_chains()
            # The sampler kernels take the same decisions but round differently (include/carma_mi355.h): bit equality is
            # the test only when the blocks and the one-GPU run were on the SAME kernel; otherwise the comparison is to
            # rounding (same decisions => the same states to 1e-6) and says so.
            kern_blocks, kern_one = [p_[3] for p_ in parts], one.pt_kernel()
            same_kernel = all(k_ == kern_one for k_ in kern_blocks)
            gth = np.concatenate([p_[0] for p_ in parts], axis=1)
            glp = np.concatenate([p_[1] for p_ in parts], axis=1)
            if same_kernel:
                equal = bool(np.array_equal(gth, uth) and np.array_equal(glp, ulp))
            else:
                equal = bool(np.allclose(gth, uth, rtol=1e-6, atol=1e-9) and np.allclose(glp, ulp, rtol=1e-6, atol=1e-9))
            check = {
                "equals_one_gpu_run": equal,
                "comparison": "bit for bit" if same_kernel else "to 1e-6 (the runs were on different sampler kernels)",
                "kernels_compared": {"blocks": kern_blocks, "one_gpu_run": kern_one},
                "boundary_checksums_agree": [p_[2] for p_ in parts],
                "iterations_checked": WARM,
            }
            del one
    barrier()
    tl0 = time.perf_counter()
    run(iters)
    tl = time.perf_counter() - tl0
    barrier()
    rates = None
    if dist is not None:
        tt = torch.tensor([tl], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        tl = float(tt.item())
        if world > 1:
            stats = [None] * world
            dist.all_gather_object(stats, (sh4.nprop_boundary, sh4.nswap_boundary, ctx4.pt_boundary_check() if comm is not None else None))
            # a boundary is counted by both of its sides; rank r's numbers cover its one or two boundaries
            rates = [{"rank": r_, "proposed": int(s_[0]), "accepted": int(s_[1]), "rate": (s_[1] / s_[0]) if s_[0] else None,
                      "checksums_agree": s_[2]} for r_, s_ in enumerate(stats)]
    res = {
        "metric": "MCMC iterations/s of ONE temperature ladder sharded across the ranks",
        "iters_per_s": iters / tl, "chain_evals_per_s": TG * R4 * iters / tl,
        "temperatures": TG, "temperatures_per_rank": TG // world, "replicas": R4, "iters": iters,
        "scaling": "strong",
        "transport": ("rccl send/recv of %d doubles per boundary and iteration" % (R4 * 17 + 1)) if comm is not None
        else ("torch.distributed stand-in (ranks share a GPU)" if world > 1 else "none (one block)"),
        "rccl_ranks": comm.size if comm is not None else 1,
        "sweep": "hot -> cold over every adjacent pair every iteration, block boundaries included (the one-GPU sampler's sweep)",
        "boundary_swaps_by_rank": rates,
        "first_contact_check": check,
        "config": "configs[3] shape: CARMA(7,6), n=10000 (0.1+|Cauchy| steps), 8 temperatures x %d replicas" % R4,
    }
    if comm is not None:
        comm.close()
    return res


def _host_cpus():
    """CPUs this process may actually use: the affinity mask, cut by the cgroup CPU quota when there is one."""
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:             # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = f.read().split()
            if q != "max":
                quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, per = float(f.read()), float(g.read())
                if q > 0:
                    quota = q / per
        except (OSError, ValueError):
            pass
    model = None
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return aff, quota, model


def cpu_baseline(t, y, yerr, p, q, max_stdev, theta, budget_s):
    """The CPU comparator of SURVEY.md 8(d): the oracle's restatement of kfilter.cpp + carpack LogDensity (the
    reference's own C++ is unbuildable here) built -O3 -march=native ON THIS BOX, timed on its host cores on a bounded
    sample of the same workload: one thread and all usable cores (OpenMP over evaluations), median of >= 3 repetitions."""
    import oracle as orc
    # two builds of the same source, both -march=native without fast-math: -O3 (as SURVEY 8d words it) and -O2 -- gcc's -O3
    # vectoriser pessimises the short complex loops of this code by 3x.  The comparator is the FASTER of the two.
    cands = {}
    for variant in ("O3", "O2"):
        mv = orc.NativeComparator(t, y, yerr, p, q, max_stdev, variant=variant)
        mv.logdensity_batch(theta[:64], nthreads=1)
        t0 = time.perf_counter()
        mv.logdensity_batch(theta[:256], nthreads=1)
        cands[variant] = (256 / (time.perf_counter() - t0), mv)
    best = max(cands, key=lambda k: cands[k][0])
    m = cands[best][1]
    aff, quota, model = _host_cpus()
    cores = max(1, min(aff, m.max_threads(), int(quota) if quota and quota >= 1 else aff))
    B = theta.shape[0]

    def rate(batch, nthreads, seconds, min_reps=3):
        m.logdensity_batch(batch[: max(64, batch.shape[0] // 8)], nthreads=nthreads)      # warm the threads up
        rates = []
        t_end = time.perf_counter() + seconds
        while len(rates) < min_reps or time.perf_counter() < t_end:
            t0 = time.perf_counter()
            m.logdensity_batch(batch, nthreads=nthreads)
            rates.append(batch.shape[0] / (time.perf_counter() - t0))
            if len(rates) >= 200:
                break
        return float(np.median(rates)), len(rates)

    v1, reps1 = rate(theta, 1, budget_s * 0.25)
    # all cores: a tile of ~1024 evaluations per thread so that OpenMP start-up and the static schedule's tail are amortised
    big = np.tile(theta, (max(1, cores * 1024 // B), 1))
    vN, repsN = rate(big, cores, budget_s * 0.45)
    # where the scaling goes: the same tile on fewer threads (median of 3 each)
    scaling = {}
    for nt in sorted({2, 4, 8, 16, 32, 64, cores // 2}):
        if 1 < nt < cores:
            scaling[str(nt)] = rate(np.tile(theta, (max(1, nt * 1024 // B), 1)), nt, 0.0)[0]
    scaling["1"], scaling[str(cores)] = v1, vN
    return {
        "value": vN,
        "unit": "evals/s",
        "cores": cores,
        "kind": "port",
        "build": "gcc %s -fopenmp (no fast-math), built on this box -- the faster of the -O3 and -O2 builds (one-thread probe: "
                 "%s); the parity oracle is a separate -O2 -ffp-contract=off build of the same source" % (
                     orc.NativeComparator.FLAGS[best], ", ".join("%s %.0f evals/s" % (orc.NativeComparator.FLAGS[k], cands[k][0]) for k in cands)),
        "cpu_model": model,
        "cpus_in_affinity_mask": aff,
        "cgroup_cpu_quota": quota,
        "os_cpu_count": os.cpu_count(),
        "sample": "the step's own 1024-theta batch (CARMA(5,3), n=270): %d x %d evals on %d OpenMP threads, "
                  "%d x %d evals on one thread; medians" % (repsN, big.shape[0], cores, reps1, B),
        "single_thread_value": v1,
        "parallel_efficiency": vN / v1 / cores,
        "thread_scaling": scaling,
    }


if __name__ == "__main__":
    main()
