#!/bin/bash
# Run ON THE GPU BOX: PMC counters of k_logdens_carma_lpc<5,3> (16 384 evaluations: one workgroup per CU) next to the plain
# lane kernel on the same launch (CARMA_TUNE_LPC_MAX=0), tools/tput_variant.py's first size.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
REPO=$PWD
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/pmc_lpc
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/lpc -o t -- python3 $REPO/tools/tput_variant.py > $OUT/lpc.log 2>&1
echo "rc=$?"
export CARMA_TUNE_LPC_MAX=0 CARMA_TUNE_LANE_MIN=0
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/lane -o t -- python3 $REPO/tools/tput_variant.py > $OUT/lane.log 2>&1
echo "rc=$?"
python3 - <<PY
import csv, glob
for d in ("lpc", "lane"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = {}
        for row in csv.DictReader(open(f)):
            if "k_logdens_carma_l" not in row["Kernel_Name"]:
                continue
            key = (row["Kernel_Name"].split("(")[0][-32:], row["Grid_Size"])
            acc.setdefault(key, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
        for key, cs in sorted(acc.items()):
            print(key[0], "grid", key[1], " ".join("%s %.4g" % (k, sum(v) / len(v)) for k, v in sorted(cs.items())), "(%d dispatches)" % len(next(iter(cs.values()))))
PY
