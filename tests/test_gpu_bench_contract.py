"""The bench.py contract on the GPU box: one JSON line with the required keys at N=1, and the N>1 launch
path (torch.distributed.run, one rank per process, barrier + max-over-ranks) exercised with two ranks
sharing cuda:0 (CARMA_BENCH_SHARE_GPU=1: gloo for the barrier, since RCCL refuses two ranks on one device)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline"}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _json_line(out):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "50", "--warmup", "5", "--mcmc-iters", "200",
                        "--cpu-seconds", "1", "--ladder-iters", "20", "--api-scale", "0.02"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _json_line(r.stdout)
    assert KEYS | {"cpu_baseline"} <= set(j)
    # the reference's own workloads through the drop-in Python API (round 6; scaled down here)
    assert "api_legs_error" not in j, j.get("api_legs_error")
    for k, call in (("quickstart_ogle", "p=6, q=0"), ("readme_run", "p=5, q=3")):
        a = j[k]
        assert call in a["call"] and a["wall_s"] > 0 and a["logpost_finite"] and a["temperatures"] == 10 and a["samples"] >= 200
        assert set(a["split_s"]) == {"sampler_call", "post_hoc_loglik_batch", "sigma_noise_batch", "carma_sample_rest"}
        assert abs(sum(a["split_s"].values()) - a["wall_s"]) < 1e-6 * max(1.0, a["wall_s"]) and min(a["split_s"].values()) >= 0.0
        assert a["cpu_port_single_thread_estimate_s"] > 0
    assert j["choose_order"]["orders"] == 28 and j["choose_order"]["wall_s"] > 0 and 1 <= j["choose_order"]["chosen"][0] <= 7
    assert j["n_gpus"] == 1 and j["steps"] == 50 and j["unit"] == "evals/s" and j["dtype"] == "f64"
    assert j["finite_in_last_batch"] == 1024 and j["value"] > 1e6           # north_star target on one GPU
    rf, cb = j["roofline"], j["cpu_baseline"]
    assert rf["bound"] == "hbm" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1
    assert abs(j["ms_per_step"] * 1e-3 * j["value"] - 1024) < 1e-6 * 1024
    # the kernel label comes from the library's own launch table, the counters from a profile of that very kernel
    assert rf["kernel"] == "k_logdens_carma_w2<5>" and rf["binding_resource"] == "fp64_valu_issue"      # (round 6: the two-sided windowed pipeline)
    assert rf["traffic"] is None or (rf["traffic_source"] and rf["traffic"] < rf["algorithmic_bytes_per_launch"])
    # no fraction on the line may exceed 1, and counters are attached only when they were measured on the running build
    def fracs(o, path=""):
        if isinstance(o, dict):
            for k, v in o.items():
                if k == "frac" or k.endswith("_frac"):
                    assert v is None or 0.0 <= v <= 1.0, (path + "/" + k, v)
                fracs(v, path + "/" + k)
    fracs(j)
    # the binding ceilings as scalars next to `frac` (None only when no counter record of this build is committed), and the
    # 2^20-evaluation leg
    for blk in (rf, j["throughput"], j["throughput_1m"], j["mcmc"]):
        assert {"valu_issue_frac", "fp64_frac", "fp64_tflops"} <= set(blk)
        for k in ("valu_issue_frac", "fp64_frac"):
            assert blk[k] is None or 0.0 < blk[k] <= 1.0, (k, blk[k])
    st = j["steady_state"]
    # (lower bound loose on purpose: inside the full suite this process is the second one with a context on the GPU -- the suite's own,
    # idle -- and its 27 ms of back-to-back launches then measured 61 us per launch twice out of two runs, against 27.4 us in three
    # runs of the same command on a box of its own: profiles/r05/README.md)
    assert st["launches"] == 1000 and 0.3 * j["value"] < st["evals_per_s"] < 1.3 * j["value"]
    t1 = j["throughput_1m"]
    assert t1["batch_per_gpu"] == 1 << 20 and t1["finite"] > 0.5 * (1 << 20) and t1["evals_per_s"] > j["throughput"]["evals_per_s"]
    assert j["build"]["build_id"] and j["rccl_first_contact"]["verdict"] == "one rank: no exchange"
    assert len(j["ranks"]) == 1 and j["ranks"][0]["device_ordinal"] == 0
    for blk in (rf["pmc_per_launch"], j["mcmc"]["pmc_per_iteration"], j["throughput"]["pmc_per_launch"]):
        assert blk is None or "SQ_INSTS_VALU" not in blk or blk["measured_on"]["source_id"] == j["build"]["source_id"] \
            or blk["measured_on"]["build_id"] == j["build"]["build_id"]
    tp, ld = j["throughput"], j["ladder_sharded"]
    assert tp["batch_per_gpu"] == 65536 and tp["kernel"] == "k_logdens_carma_lane<5>" and tp["evals_per_s"] > j["value"]
    assert ld["temperatures"] == 8 and ld["replicas"] == 128 and ld["rccl_ranks"] == 1 and ld["iters_per_s"] > 0


def test_two_ranks_sharing_the_gpu():
    env = dict(os.environ, CARMA_BENCH_SHARE_GPU="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "50", "--warmup", "5", "--mcmc-iters", "200", "--ladder-iters", "6"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _json_line(r.stdout)
    assert KEYS <= set(j) and "cpu_baseline" not in j                      # CPU leg runs on rank 0 at N=1 only
    assert [r_["rank"] for r_ in j["ranks"]] == [0, 1] and j["rccl_first_contact"]["verdict"] == "ok", j["rccl_first_contact"]
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["finite_in_last_batch"] == 1024
    assert abs(j["ms_per_step"] * 1e-3 * j["value"] - 2 * 1024) < 1e-6 * 2048          # whole-job aggregate
    ld = j["ladder_sharded"]                                                 # one ladder of 8 temperatures, 4 per rank
    assert ld["temperatures_per_rank"] == 4 and ld["scaling"] == "strong", ld
    rates = ld["boundary_swaps_by_rank"]
    assert len(rates) == 2 and rates[0]["proposed"] == rates[1]["proposed"] == 128 * 16 and rates[0]["accepted"] == rates[1]["accepted"]
    assert 0.0 < rates[0]["rate"] < 1.0
    # first contact: the two blocks' chains after ten iterations ARE the one-GPU ladder's
    assert ld["first_contact_check"]["equals_one_gpu_run"] is True, ld["first_contact_check"]


def _native_ladder_ranks(nranks):
    """The line the driver's N > 1 runs produce, with the ladder leg on the library's NATIVE path (carma_pt_iterate_sharded:
    pack kernel -> send/recv -> swap kernel on the sampler's stream, boundary checksums) -- `nranks` processes, 8 / nranks
    temperatures each.  One GPU here, so the ranks share it and the eight RCCL entry points are the shared-memory test double of
    tests/shm_transport (CARMA_RCCL_LIB); everything above them is what runs on that many GPUs."""
    import subprocess as sp
    here = os.path.join(ROOT, "tests", "shm_transport")
    lib = os.path.join(here, "libshm_rccl.so")
    sp.run(["/opt/rocm/bin/hipcc", "-O1", "-shared", "-fPIC", "-o", lib, os.path.join(here, "shm_rccl.cpp"), "-lrt", "-lpthread"],
           check=True, stdout=sp.PIPE, stderr=sp.STDOUT, timeout=600)
    env = dict(os.environ, CARMA_BENCH_SHARE_GPU="1", CARMA_RCCL_LIB=lib)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks),
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
                        "--gpus", str(nranks), "--steps", "50", "--warmup", "5", "--mcmc-iters", "100", "--no-pipelined", "--no-throughput",
                        "--ladder-iters", "6"], capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _json_line(r.stdout)
    assert j["n_gpus"] == nranks and j["finite_in_last_batch"] == 1024
    ld = j["ladder_sharded"]
    assert ld["rccl_ranks"] == nranks and ld["temperatures_per_rank"] == 8 // nranks and ld["transport"].startswith("rccl send/recv"), ld
    fc = ld["first_contact_check"]
    assert fc["equals_one_gpu_run"] is True and fc["boundary_checksums_agree"] == [1] * nranks, fc
    rates = ld["boundary_swaps_by_rank"]
    assert [r_["proposed"] for r_ in rates] == [128 * 16] + [2 * 128 * 16] * (nranks - 2) + [128 * 16]
    assert all(r_["checksums_agree"] == 1 and 0.0 < r_["rate"] < 1.0 for r_ in rates)
    assert j.get("ladder_leg_hung") in (None, False)


def test_four_ranks_with_the_native_ladder_path():
    _native_ladder_ranks(4)


def test_eight_ranks_one_temperature_each():
    """BASELINE configs[3] at its real partition -- eight ranks, ONE temperature per rank (every chain of a block is a boundary
    chain on both sides) -- so that the driver's first 8-GPU run exercises no rank count the code has never seen."""
    _native_ladder_ranks(8)


def test_ladder_leg_cannot_hold_the_line_back():
    """The ladder-sharded leg is the only one with an exchange between the ranks; a collective that never returns must not
    cost the run its JSON line: with a watchdog of 10 ms rank 0 prints the line without the leg -- and every rank leaves with
    a NON-ZERO status (a wedged exchange is a failed run to the launcher; the line is there to be read on rc != 0)."""
    env = dict(os.environ, CARMA_BENCH_SHARE_GPU="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "50", "--warmup", "5", "--no-mcmc", "--no-pipelined", "--no-throughput",
                        "--ladder-iters", "200", "--ladder-timeout", "0.01"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode != 0, "a hung ladder leg must not look like a successful run"
    assert "leaving with status 3" in r.stderr, r.stderr[-2000:]
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 2 and j["value"] > 1e6 and "no result after" in j["ladder_sharded"]["error"] and j["ladder_leg_hung"] is True
