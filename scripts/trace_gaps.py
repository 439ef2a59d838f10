#!/usr/bin/env python3
"""Idle gaps (> 15 us) between consecutive kernels, per replay, in a rocprofv3 kernel trace CSV (replays behind the longest gap).
usage: scripts/trace_gaps.py <kernel_trace.csv> <replays>"""
import csv
import sys

from trace_summary_names import family  # noqa: E402

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2])
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?"))
            for r in rows)
gaps = [(ev[i + 1][0] - ev[i][1], i) for i in range(len(ev) - 1)]
ev = ev[max(gaps)[1] + 1:]
per = len(ev) // n
print("queues", sorted(set(e[3] for e in ev)), "streams", sorted(set(e[4] for e in ev)))
for r in range(n):
    seg = ev[r * per:(r + 1) * per]
    tot = 0.0
    out = []
    for i in range(1, len(seg)):
        g = (seg[i][0] - seg[i - 1][1]) / 1e3
        if g > 2.0:
            tot += g
        if g > 15.0:
            out.append("#%d %.0fus(%s)" % (i, g, family(seg[i][2])[:18]))
    print("replay %d: wall %.2f ms, gaps %.0f us: %s" % (r, (seg[-1][1] - seg[0][0]) / 1e6, tot, " ".join(out[:14])))
