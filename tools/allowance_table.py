#!/usr/bin/env python3
"""Markdown table of the parity allowances a GPU test run CONSUMED (tests/helpers.py record_allowance ->
gpurun_out/parity_allowances.json, written by tests/conftest.py at the end of a pytest session).

    python tools/allowance_table.py [gpurun_out/parity_allowances.json] > table.md

One row per (clause, test): entries that used the clause / entries the clause would have allowed / population, and the worst
distance of the device and of the oracle from the quad-precision value among them.  DESIGN.md section 4 carries the round's
table; the JSON is committed as profiles/rNN/parity_allowances.json."""
import json
import os
import sys

path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_allowances.json")
recs = json.load(open(path))["records"]
rows = {}
for r in recs:
    key = (r["clause"], r["test"].split("::")[-1] if r["test"] else r["what"])
    a = rows.setdefault(key, dict(used=0, allowed=0, pop=0, wd=0.0, wo=0.0, n=0))
    a["used"] += r["used"]
    a["allowed"] += r["allowed"]
    a["pop"] += r["population"] or 0
    a["wd"] = max(a["wd"], r["worst_device"] or 0.0)
    a["wo"] = max(a["wo"], r["worst_oracle"] or 0.0)
    a["n"] += 1
print("| clause | test | used | allowed | of | worst device | worst oracle |")
print("|---|---|---|---|---|---|---|")
for (clause, test), a in sorted(rows.items()):
    print("| %s | `%s`%s | %d | %d | %d | %s | %s |" % (clause, test, " (%d calls)" % a["n"] if a["n"] > 1 else "", a["used"], a["allowed"], a["pop"],
                                                       "%.1e" % a["wd"] if a["wd"] else "–", "%.1e" % a["wo"] if a["wo"] else "–"))
tot = sum(a["used"] for a in rows.values())
print("\n%d uses of an allowance in %d (clause, test) pairs; a run in which a clause is not used leaves no row." % (tot, len(rows)))
