#!/usr/bin/env python3
"""rocprofv3 --pmc counter_collection.csv -> per-kernel mean counter values (CSV under profiles/).
usage: scripts/pmc_summary.py out.csv counter_collection.csv [more.csv ...]"""
import collections
import csv
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for path in sys.argv[2:]:
    with open(path) as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            a = acc[k][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
counters = sorted({c for k in acc for c in acc[k]})
with open(sys.argv[1], "w") as f:
    f.write("# source: rocprofv3 --pmc <counter> (one pass per TCC counter); mean per dispatch\n")
    f.write("kernel,dispatches," + ",".join(counters) + "\n")
    rows = []
    for k, d in acc.items():
        n = max(v[1] for v in d.values())
        rows.append((sum(v[0] for v in d.values()), k, n, d))
    for _, k, n, d in sorted(rows, reverse=True, key=lambda t: t[0]):
        f.write("%s,%d,%s\n" % (k[:110].replace(",", ";"), n,
                                ",".join("%.1f" % (d[c][0] / d[c][1]) if c in d else "" for c in counters)))
print("wrote", sys.argv[1], len(acc), "kernels")
