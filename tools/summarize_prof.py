#!/usr/bin/env python3
"""Condenses the rocprofv3 output of tools/profile_round.sh into the files committed under profiles/:
   kernel_stats_<tag>.csv       (the --stats kernel summary, verbatim)
   pmc_<tag>.json               headline kernel k_logdens_carma_w2<5> (round 6: two-sided; round 5: k_logdens_carma_w<5>; before: k_logdens_carma_p3l<5>):
                                per-LAUNCH mean/min/max of each counter
   pmc_<tag>_ptrow.json         sampler kernel k_pt_row<5,...>: counters summed over its dispatches / iterations run
                                (the bench's sampler leg under --mcmc-iters 200: 100 warm-up + 200 timed = 300) -> per ITERATION
   pmc_<tag>_tput.json          throughput kernel k_logdens_carma_lane<5> (one evaluation per lane): per launch of 65 536 evaluations
   pmc_<tag>_tput1m.json        the same kernel's launches of 2^20 evaluations (the throughput_1m leg)
usage: summarize_prof.py <tag> [outdir] [sampler-iterations-in-the-pmc-run (300)]"""
import csv, glob, json, os, shutil, sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(root, "profiles", "r06")
pt_iters = int(sys.argv[3]) if len(sys.argv) > 3 else 300
os.makedirs(dst, exist_ok=True)
sys.path.insert(0, root)
from carma_pack_amd._lib import build_ids  # noqa: E402  (which build these counters belong to: bench.py checks it)

IDS = build_ids()
if os.path.exists(os.path.join(src, "ids.json")):           # stamped on the GPU box at measurement time
    IDS = json.load(open(os.path.join(src, "ids.json")))
for f in glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(dst, "kernel_stats_%s.csv" % tag))
    print("copied", f)
NOTES = [
    "bench.py --steps 50 --warmup 5 --no-cpu --no-pipelined --no-ladder --no-mcmc-large --no-api --mcmc-iters 200 under rocprofv3 --pmc <counters> --kernel-trace, "
    "one counter set per run (tools/profile_round.sh)",
    "FETCH_SIZE/WRITE_SIZE are KiB; gfx950 FETCH_SIZE may under-count wide coalesced reads by 2x "
    "(MI355X_MICROARCH.md): upper bound on HBM reads = 2*FETCH_SIZE",
    "SQ_WAVE_CYCLES and SQ_ACTIVE_INST_VALU count quad-cycles",
]


def collect(kern, grid=None, per_iteration=0):
    res, disp = {}, None
    for f in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        vals = {}
        for row in csv.DictReader(open(f)):
            if kern not in row["Kernel_Name"].replace(" ", "") or (grid is not None and row["Grid_Size"] != str(grid)):
                continue
            vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
            disp = {k: row[k] for k in ("Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size",
                                        "VGPR_Count", "SGPR_Count")}
        for c, v in vals.items():
            if per_iteration:
                res[c] = {"dispatches": len(v), "mean": sum(v) / per_iteration, "total": sum(v)}
            else:
                res[c] = {"dispatches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)}
    res["_dispatch"] = disp
    res["_per"] = ("sampler iteration (counters summed over the kernel's dispatches / %d iterations)" % per_iteration) if per_iteration else "launch"
    res["_notes"] = NOTES
    res["_ids"] = IDS
    return res


for name, kern, grid, it in (("", "k_logdens_carma_w2<5,", 131072, 0), ("_ptrow", "k_pt_row<5,", None, pt_iters),      # (1024 evaluations: 512 workgroups x 256 threads)
                             ("_tput", "k_logdens_carma_lane<5", 65536, 0), ("_tput1m", "k_logdens_carma_lane<5", 1 << 20, 0)):
    # (the 65 536-evaluation launches and, separately, the 2^20 ones of throughput_1m)
    r = collect(kern, grid, it)
    if r["_dispatch"] is None:
        print("no dispatches of", kern)
        continue
    out = os.path.join(dst, "pmc_%s%s.json" % (tag, name))
    json.dump(r, open(out, "w"), indent=1)
    print("wrote", out, sorted(k for k in r if not k.startswith("_")))
