#!/usr/bin/env python3
"""Every idle gap > N us in the tail of a rocprofv3 kernel trace CSV: where it is (kernels since the previous big gap),
the kernels on both sides.  usage: scripts/trace_gaps_all.py <kernel_trace.csv> [skip_fraction=0.5] [min_us=15]"""
import csv
import sys

from trace_summary_names import family  # noqa: E402

rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 15.0
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
ev = ev[int(len(ev) * skip):]
since, tot, wall = 0, 0.0, (ev[-1][1] - ev[0][0]) / 1e3
for i in range(1, len(ev)):
    g = (ev[i][0] - ev[i - 1][1]) / 1e3
    since += 1
    if g > 2.0:
        tot += g
    if g > min_us:
        print("t=%9.1f us  gap %6.1f us  after %5d kernels   %s -> %s" % ((ev[i][0] - ev[0][0]) / 1e3, g, since,
                                                                          family(ev[i - 1][2])[:28], family(ev[i][2])[:28]))
        since = 0
print("window %.1f ms, gaps > 2 us: %.1f us (%.2f %%)" % (wall / 1e3, tot, 100 * tot / wall))
