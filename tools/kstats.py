#!/usr/bin/env python3
"""Prints calls / average / total per kernel from a rocprofv3 kernel_stats CSV."""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Name"].split("(")[0].replace("plk::", "").replace("void ", "")
    print(f'{name:34s} calls={r["Calls"]:>6s} avg_us={float(r["AverageNs"])/1e3:10.2f} total_ms={float(r["TotalDurationNs"])/1e6:10.3f} pct={r["Percentage"]}')
