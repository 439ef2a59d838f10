#!/usr/bin/env python3
"""Kernel trace CSV -> launches whose grid is smaller than the chip but which take long: name, workgroups, average us, calls.
usage: scripts/small_grid_census.py <kernel_trace.csv> [max_workgroups=256] [min_us=8]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
max_wg = int(sys.argv[2]) if len(sys.argv) > 2 else 256
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 8.0
ev = sorted(rows, key=lambda r: int(r["Start_Timestamp"]))
ev = ev[len(ev) // 2:]                      # second half of the run: the replays
agg = collections.defaultdict(lambda: [0, 0.0])
for r in ev:
    wg = 1
    for ax in "XYZ":
        wg *= max(1, int(r["Grid_Size_" + ax]) // max(1, int(r["Workgroup_Size_" + ax])))
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
    a = agg[(name, wg)]
    a[0] += 1
    a[1] += us
out = [(t, n, name, wg) for (name, wg), (n, t) in agg.items() if wg <= max_wg and t / n >= min_us]
tot = sum(t for t, *_ in out)
print("launches with <= %d workgroups and >= %.0f us: %.2f ms of the analysed half" % (max_wg, min_us, tot / 1e3))
for t, n, name, wg in sorted(out, reverse=True)[:60]:
    print("%8.1f us total  %5d calls  %7.1f us avg  %6d wgs  %s" % (t, n, t / n, wg, name))
