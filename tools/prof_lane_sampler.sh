#!/bin/bash
# Per-kernel times of one large-ensemble sampler run (rocprofv3 --kernel-trace --stats): tools/prof_lane_sampler.sh [R] [kernel]
# (on the GPU box; writes gpurun_out/lane_sampler_kernels.txt)
R=${1:-4096}; K=${2:-lane}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ml
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ml -o ml -- python3 "$ROOT/tools/mcmc_lane_probe.py" "$K" "$R" 2>&1 | grep "kernel " > "$ROOT/gpurun_out/lane_sampler_kernels.txt"
python3 - "$ROOT" <<'PY'
import csv, glob, sys
fs = glob.glob("/tmp/ml/**/*kernel_stats.csv", recursive=True)
out = open(sys.argv[1] + "/gpurun_out/lane_sampler_kernels.txt", "a")
if not fs:
    out.write("no kernel_stats.csv under /tmp/ml\n")
for f in fs[:1]:
    for r in csv.DictReader(open(f)):
        out.write("%-72s calls %6s  avg %10.1f us  %5s %%\n" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
cat "$ROOT/gpurun_out/lane_sampler_kernels.txt"
