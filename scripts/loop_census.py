#!/usr/bin/env python3
"""Instruction census of the main loop (largest backward-branch span) of one kernel in a gfx950 assembly listing.
usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o k.s file.hip; scripts/loop_census.py k.s <substring of the kernel symbol> [-v]"""
import collections
import re
import sys

txt = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = next(i for i, l in enumerate(txt) if re.match(r"^_Z\S*%s\S*:" % re.escape(key), l))
end = next(i for i in range(start, len(txt)) if txt[i].startswith(".Lfunc_end"))
lines = [l.strip() for l in txt[start:end]]
labels = {}
for i, l in enumerate(lines):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = i
best = None
for i, l in enumerate(lines):
    m = re.match(r"^s_cbranch_\w+ (\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        span = (labels[m.group(1)], i)
        if best is None or span[1] - span[0] > best[1] - best[0]:
            best = span
cnt = collections.Counter()
for l in lines[best[0]:best[1]]:
    if not l or l[0] in ".;/":
        continue
    cnt[l.split()[0]] += 1
groups = collections.Counter()
for op, c in cnt.items():
    g = ("mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else
         "vmem" if op.startswith(("buffer_", "global_")) else "salu" if op.startswith("s_") else "other")
    groups[g] += c
print("loop of %d instructions:" % sum(cnt.values()), dict(groups))
for op, c in cnt.most_common(60 if "-v" in sys.argv else 25):
    print("  %-28s %d" % (op, c))
