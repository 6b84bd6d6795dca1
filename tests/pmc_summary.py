"""Summarise rocprofv3 --pmc CSV output (counter_collection.csv files under a directory): per
kernel name, the mean of every counter over its dispatches.  With --json also prints one JSON
object {kernel: {counter: mean, "dispatches": n}}.  Usage: python3 tests/pmc_summary.py <dir> [--json]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not any(t in k for t in ("convgemm", "resconv", "attention", "gn_glu", "preproc", "energy")):
            continue
        short = k.split("(")[0].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
        acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in acc.items():
    print(k)
    out[k] = {}
    for c, v in sorted(cs.items()):
        print(f"    {c:32s} {sum(v) / len(v):16.1f}  (n={len(v)})")
        out[k][c] = sum(v) / len(v)
        out[k]["dispatches"] = len(v)
if "--json" in sys.argv:
    print(json.dumps(out))
