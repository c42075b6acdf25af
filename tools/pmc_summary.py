#!/usr/bin/env python3
"""Averages rocprofv3 --pmc counter values per kernel name from a counter_collection CSV."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("plk::", "").replace("void ", "")
    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in sorted(acc.items()):
    print(name, " ".join(f"{c}={sum(v)/len(v):.4g}(n={len(v)})" for c, v in sorted(cs.items())))
