#!/usr/bin/env python3
"""Per-kernel resource table (VGPR / AGPR / scratch / occupancy / LDS) of one .hip file.
usage: scripts/kres.py file.hip [extra hipcc flags]"""
import re
import subprocess
import sys

src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/kres.o"] + sys.argv[2:]
out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, cwd="/tmp").stdout
rows, cur = [], None
for line in out.splitlines():
    if "error" in line:
        print(line)
    m = re.search(r"remark: (.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip().replace("_ZN12_GLOBAL__N_1", "")[:70]}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.rsplit(":", 1)
        cur[k.strip()] = v.strip()
seen = set()
for r in rows:
    if r["name"] in seen:
        continue
    seen.add(r["name"])
    print("%-72s vgpr=%s agpr=%s scratch=%s occ=%s lds=%s" % (
        r["name"], r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize [bytes/lane]"),
        r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))
