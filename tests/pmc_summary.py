#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output (counter_collection.csv files under a directory):
per kernel name, the mean of every counter over its dispatches."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "convgemm" not in k and "resconv" not in k and "attention" not in k:
            continue
        short = k.split("(")[0].replace("void (anonymous namespace)::", "")
        acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"    {c:32s} {sum(v) / len(v):16.1f}  (n={len(v)})")
