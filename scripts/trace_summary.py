#!/usr/bin/env python3
"""Per-iteration summary of a rocprofv3 kernel trace CSV: GPU busy / idle and time by kernel family.
usage: scripts/trace_summary.py <kernel_trace.csv> <iterations in trace> [skip_fraction_at_start | --after-gap]
--after-gap [ms]: keep only the kernels behind the last idle gap of at least that many ms (scripts/phase_trace.py sleeps there)"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
iters = float(sys.argv[2])
after_gap = len(sys.argv) > 3 and sys.argv[3] == "--after-gap"
skip = float(sys.argv[3]) if len(sys.argv) > 3 and not after_gap else 0.0
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
ev = ev[int(len(ev) * skip):]
if after_gap:
    # the LAST idle gap of at least `--after-gap <ms>` (default 400): the probe sleeps 1 s right before the replays it wants
    # summarised; a longer stall earlier in the run (graph instantiation on a cold box) must not win
    min_gap = float(sys.argv[4]) * 1e6 if len(sys.argv) > 4 else 400e6
    gaps = [(ev[i + 1][0] - ev[i][1], i) for i in range(len(ev) - 1)]
    long_gaps = [i for gap, i in gaps if gap >= min_gap]
    ev = ev[(long_gaps[-1] if long_gaps else max(gaps)[1]) + 1:]
wall = ev[-1][1] - ev[0][0]
busy, cur = 0, ev[0][0]
for s, e, _ in ev:
    busy += max(0, e - max(s, cur))
    cur = max(cur, e)
print("kernels/iter %.0f   wall %.2f ms/iter   GPU busy %.2f ms/iter (%.1f%%)   idle %.1f%%"
      % (len(ev) / iters, wall / 1e6 / iters, busy / 1e6 / iters, 100.0 * busy / wall, 100.0 * (1 - busy / wall)))


from trace_summary_names import family  # noqa: E402



agg = collections.defaultdict(lambda: [0, 0])
for s, e, n in ev:
    a = agg[family(n)]
    a[0] += 1
    a[1] += e - s
tot = sum(a[1] for a in agg.values())
small = sum(1 for s, e, _ in ev if e - s < 10000)
small_t = sum(e - s for s, e, _ in ev if e - s < 10000)
print("kernels < 10 us: %.0f/iter, %.2f ms/iter" % (small / iters, small_t / 1e6 / iters))
print("%-62s %9s %10s %9s" % ("kernel", "calls/it", "ms/iter", "avg us"))
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[4]) if len(sys.argv) > 4 else 45]:
    print("%-62s %9.1f %10.3f %9.1f" % (n, c / iters, t / 1e6 / iters, t / 1e3 / c))
