#!/usr/bin/env python3
"""Kernel SEQUENCE of the last replay in a rocprofv3 kernel trace CSV (start offset, duration, gap to the previous kernel,
short name) — the chains of tiny launches that the per-family table of scripts/trace_summary.py hides.
usage: scripts/trace_sequence.py <kernel_trace.csv> <replays behind the longest gap>"""
import csv
import sys

from trace_summary_names import family  # noqa: E402  (same short names as trace_summary.py)

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2])
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
gaps = [(ev[i + 1][0] - ev[i][1], i) for i in range(len(ev) - 1)]
ev = ev[max(gaps)[1] + 1:]
per = len(ev) // n
ev = ev[-per:]
t0, prev = ev[0][0], ev[0][0]
for s, e, name in ev:
    print("%9.1f %8.1f %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, family(name)))
    prev = e
