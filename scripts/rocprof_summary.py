#!/usr/bin/env python3
"""rocprofv3 (rocpd sqlite) -> per-kernel summary CSV kept under profiles/.
usage: scripts/rocprof_summary.py results.db out.csv [note]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
with open(sys.argv[2], "w") as f:
    if len(sys.argv) > 3:
        f.write("# %s\n" % sys.argv[3])
    f.write("# source: rocprofv3 --kernel-trace --stats ; durations in microseconds\n")
    f.write("kernel,calls,total_us,avg_us,percent\n")
    for name, calls, tot, avg, pct in rows:
        short = name.replace("(anonymous namespace)::", "").replace("void ", "")
        short = short.split("(")[0][:110].replace(",", ";")
        f.write("%s,%d,%.1f,%.2f,%.2f\n" % (short, calls, tot, avg, pct))
print("wrote", sys.argv[2], len(rows), "kernels")
