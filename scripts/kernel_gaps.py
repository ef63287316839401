#!/usr/bin/env python3
"""Developer aid: idle time between consecutive kernels of a substep, from a rocprofv3 --kernel-trace CSV.
usage: kernel_gaps.py <kernel_trace.csv>"""
import csv, sys, collections
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-60:]))
rows.sort()
rows = rows[len(rows) // 2:]          # steady state
gaps = collections.defaultdict(list)
durs = collections.defaultdict(list)
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    gaps[(n0[:40], n1[:40])].append(s1 - e0)
    durs[n0[:40]].append(e0 - s0)
for k, v in sorted(durs.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print("dur  %-42s n=%5d median %.2f us  total %.1f us" % (k, len(v), v[len(v) // 2] / 1e3, sum(v) / 1e3))
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print("gap  %-42s -> %-42s n=%5d median %.2f us  p90 %.2f us total %.1f us" % (k[0], k[1], len(v), v[len(v) // 2] / 1e3, v[int(len(v) * 0.9)] / 1e3, sum(v) / 1e3))
