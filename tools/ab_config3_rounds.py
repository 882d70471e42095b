#!/usr/bin/env python3
"""Round 6: BASELINE configs[3] -- CARMA(7,6), n = 10^4, 8 temperatures x 128 ladders on one GPU -- on THIS tree
and on the round-4 tree (git worktree of the round-4 final commit under build_var/r4_tree, its own library and Python package), in
alternating processes on one box.  Round 4's record had 779 it/s, rounds 5 and 6 have 752-763.
usage: ab_config3_rounds.py [other-tree (build_var/r4_tree)] [repeats (3)]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
other = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else os.path.join(ROOT, "build_var", "r4_tree")
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
CHILD = r'''
import sys, time, json
sys.path.insert(0, sys.argv[1])
import numpy as np
import carma_pack_amd as cpa
from carma_pack_amd.synth import config4_series
from carma_pack_amd import parallel as par
t, y, e, _ = config4_series(10000, seed=4)
ctx = cpa.Context(t, y, e, 7, 6)
ctx.pt_create(8, 128, 10 ** 9, seed=17, temperatures=par.ladder_temperatures(8))
ctx.pt_shard(8, 0, 0)
ctx.pt_start(None)
ctx.pt_iterate(10)
res = []
for it in (60, 60, 120):
    t0 = time.perf_counter(); ctx.pt_iterate(it); res.append(round(it / (time.perf_counter() - t0), 1))
print(json.dumps(dict(tree=sys.argv[1], lib=cpa._lib.LIB_PATH, it_per_s=res, kernel=ctx.pt_kernel())))
'''
for r in range(reps):
    for tree in (ROOT, other):
        p = subprocess.run([sys.executable, "-c", CHILD, tree], cwd=tree, capture_output=True, text=True, timeout=900)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        print(line[-1] if line else "FAILED %s: %s" % (tree, p.stderr[-400:]), flush=True)
