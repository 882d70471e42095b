#!/usr/bin/env python3
"""Lists the loops (backward branches) of one kernel in a hipcc -S listing, with instruction counts by class between the loop's
label and its backward branch.  usage: asm_loops.py listing.s mangled-name-substring"""
import re, sys, collections
lines = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and key in l and l.rstrip().split(":")[0].endswith(key) or (l.startswith("_ZN") and key in l.split(":")[0]))
end = next(i for i in range(start, len(lines)) if lines[i].startswith("\t.amdhsa_kernel") or lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
labels = {}
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB[0-9_]+):", l)
    if m:
        labels[m.group(1)] = i
def cls(op):
    if op.startswith("v_"):
        if "f64" in op or op.startswith("v_fma_f64") or op.startswith("v_rcp") : return "valu_f64"
        return "valu_other"
    if op.startswith("s_"):
        return "salu" if not op.startswith("s_waitcnt") and not op.startswith("s_nop") and not op.startswith("s_barrier") else op.split()[0]
    if op.startswith("ds_"): return "lds"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"): return "vmem"
    return "other"
for i, l in enumerate(body):
    m = re.match(r"^\s+s_cbranch_\w+\s+(\.LBB[0-9_]+)|^\s+s_branch\s+(\.LBB[0-9_]+)", l)
    if m:
        tgt = m.group(1) or m.group(2)
        if tgt in labels and labels[tgt] < i:
            cnt = collections.Counter()
            for k in range(labels[tgt], i + 1):
                t = body[k].strip()
                if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"): continue
                cnt[cls(t.split()[0])] += 1
            tot = sum(cnt.values())
            print("loop %s: lines %d..%d, %d instr: %s" % (tgt, start + labels[tgt], start + i, tot, dict(cnt)))
print("kernel lines", start, end, "total instr", sum(1 for l in body if l.startswith("\t") and not l.strip().startswith(".") and not l.strip().startswith(";")))
