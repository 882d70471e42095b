#!/usr/bin/env python3
"""Condenses the rocprofv3 output of tools/profile_round.sh into the files committed under profiles/:
   kernel_stats_<tag>.csv  (the --stats kernel summary, verbatim)
   pmc_<tag>.json          (per-dispatch mean/min/max of each counter for the dominant bench kernel)
usage: summarize_prof.py <tag> <kernel-substring> [outdir]"""
import csv, glob, json, os, shutil, sys

tag, kern = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = sys.argv[3] if len(sys.argv) > 3 else os.path.join(root, "profiles", "r02")
os.makedirs(dst, exist_ok=True)
for f in glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(dst, "kernel_stats_%s.csv" % tag))
    print("copied", f)
res, disp = {}, None
for f in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    vals = {}
    for row in csv.DictReader(open(f)):
        if kern not in row["Kernel_Name"]:
            continue
        vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
        disp = {k: row[k] for k in ("Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size",
                                    "VGPR_Count", "SGPR_Count")}
    for c, v in vals.items():
        res[c] = {"dispatches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)}
res["_dispatch"] = disp
res["_notes"] = [
    "bench.py --steps 50 --warmup 5 --no-cpu --no-mcmc under rocprofv3 --pmc <counter> --kernel-trace, one counter set per run",
    "FETCH_SIZE/WRITE_SIZE are KiB per dispatch; gfx950 FETCH_SIZE may under-count wide coalesced reads by 2x "
    "(MI355X_MICROARCH.md): upper bound on HBM reads = 2*FETCH_SIZE",
    "SQ_WAVE_CYCLES and SQ_ACTIVE_INST_VALU count quad-cycles",
]
out = os.path.join(dst, "pmc_%s.json" % tag)
json.dump(res, open(out, "w"), indent=1)
print("wrote", out, sorted(k for k in res if not k.startswith("_")))
