#!/bin/bash
# kernel trace of the driver-flag bench (--steps 20 --warmup 5): start/end of every p3l launch, to see where the short
# run's extra 1.4 us per step sits (slower kernels, or gaps between them)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_short
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 bench.py --steps 20 --warmup 5 --no-cpu > $OUT/bench.log 2>&1
python3 - <<'PY'
import csv, glob, os
f = glob.glob(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/trace_short/**/*kernel_trace.csv"), recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
p = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
idx = [i for i, r in enumerate(p) if "p3l<5>" in r[2] or "p3lILi5" in r[2]]
print("p3l<5> launches:", len(idx))
prev_end = None
out = []
for i in idx:
    s, e, _ = p[i]
    gap = (s - prev_end) / 1e3 if prev_end is not None else float("nan")
    out.append((e - s) / 1e3, ) if False else out.append(((e - s) / 1e3, gap))
    prev_end = e
for k, (d, g) in enumerate(out[-60:]):
    print("%3d dur %.2f us gap-before %.2f us" % (k, d, g))
PY
