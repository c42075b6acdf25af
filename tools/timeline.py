#!/usr/bin/env python3
"""Prints the kernel timeline of the last proof in a rocprofv3 kernel-trace CSV (start, end, duration, queue, kernel)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
minlen = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 3      # accumulation launches after the last proof (bench.py's isolated launches)
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('plk::', '').replace('void ', ''), r['Queue_Id']) for r in rows)
acc = [e for e in ev if 'accumulate' in e[2]]
t_begin = acc[-18 - skip][0] - 1_500_000
t_end = acc[-skip][0] - 2_000_000 if skip else ev[-1][1] + 1
sel = [e for e in ev if t_begin <= e[0] < t_end]
t0 = sel[0][0]
busy = 0; cs, ce = sel[0][0], sel[0][1]
for s, e, _, _ in sel[1:]:
    if s > ce: busy += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
busy += ce - cs
accbusy = sum(e - s for s, e, n, _ in sel if 'accumulate' in n)
print("span ms %.3f  busy-union ms %.3f  accumulate-sum ms %.3f" % ((sel[-1][1] - t0) / 1e6, busy / 1e6, accbusy / 1e6))
for s, e, n, q in sel:
    if (e - s) / 1e6 >= minlen:
        print(f"{(s-t0)/1e6:8.3f} {(e-t0)/1e6:8.3f} {(e-s)/1e6:7.3f} q{q} {n}")
