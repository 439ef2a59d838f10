#!/usr/bin/env python3
"""Per-kernel evidence table from one profiling round (scripts/profile_round.sh):
kernel | launches/step | ms/step | avg us | HBM GB/s (2*FETCH+WRITE over duration) | matrix-pipe busy %.
usage: scripts/kernel_table.py <gpurun_out/prof_TAG> <TAG> <steps incl. warm-up> > profiles/TAG_kernels.md"""
import csv
import sys

d, tag, steps = sys.argv[1], sys.argv[2], float(sys.argv[3])


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


stats = {}
for r in csv.DictReader(open("%s/trace/%s_kernel_stats.csv" % (d, tag))):
    stats[short(r["Name"])] = (int(r["Calls"]), float(r["TotalDurationNs"]), float(r["AverageNs"]))
pmc = {}
for sub in ("pmc_FETCH_SIZE", "pmc_WRITE_SIZE", "pmc_mfma"):
    acc = {}
    for r in csv.DictReader(open("%s/%s/%s_counter_collection.csv" % (d, sub, tag))):
        k = short(r["Kernel_Name"])
        a = acc.setdefault((k, r["Counter_Name"]), [0.0, 0])
        a[0] += float(r["Counter_Value"])
        a[1] += 1
    for (k, c), (v, n) in acc.items():
        pmc.setdefault(k, {})[c] = v / n
total = sum(v[1] for v in stats.values())
print("| kernel | launches/step | ms/step | avg us | HBM GB/s | MFMA pipe busy |")
print("|---|---|---|---|---|---|")
for k, (calls, tot, avg) in sorted(stats.items(), key=lambda kv: -kv[1][1]):
    if tot / total < 0.004:
        continue
    c = pmc.get(k, {})
    gbps = ""
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        gbps = "%.0f" % ((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 / avg)        # bytes / ns = GB/s
    busy = ""
    if c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) > 0 and c.get("GRBM_GUI_ACTIVE", 0) > 0:
        busy = "%.0f %%" % (100 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024))
    print("| `%s` | %.1f | %.3f | %.1f | %s | %s |" % (k[:70], calls / steps, tot / steps / 1e6, avg / 1e3, gbps, busy))
print()
print("Total kernel time %.2f ms/step.  HBM GB/s = (2*FETCH_SIZE + WRITE_SIZE) KB per dispatch / average duration"
      " (gfx950 FETCH_SIZE correction, see DESIGN.md §5); MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES /"
      " (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)." % (total / steps / 1e6))
