#!/bin/bash
# Run ON THE GPU BOX: PMC counters of the throughput-regime kernel (B = 16384 / 262144 launches of tools/tput_variant.py).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
REPO=$PWD
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/pmc_tput
mkdir -p $OUT
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/a -o t -- python3 $REPO/tools/tput_variant.py > $OUT/a.log 2>&1
echo "rc=$?"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/b -o t -- python3 $REPO/tools/tput_variant.py > $OUT/b.log 2>&1
echo "rc=$?"
python3 - <<PY
import csv, glob
for d in ("a", "b"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = {}
        for row in csv.DictReader(open(f)):
            if "k_logdens_carma<5" not in row["Kernel_Name"] or row["Grid_Size"] != str(262144 // 8 * 64):
                continue
            acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
        for k, v in sorted(acc.items()):
            print("%-24s mean %.4g over %d dispatches" % (k, sum(v) / len(v), len(v)))
PY
