#!/bin/bash
# GPU box: L2 (TCC) hit rate of the headline step's kernels — one --pmc pass, never combined with trace domains.
# usage: bash scripts/l2_hit_probe.sh   -> gpurun_out/l2_hit.txt
root=/root/repo
out=$root/gpurun_out/l2hit
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
BENCH="python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train --no-inversion --no-raster --no-pmc --no-split-bf16"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $out/p -o l2 -- $BENCH > $out/run.log 2>&1
f=$(find $out/p -name "*counter_collection.csv" | head -1)
if [ -z "$f" ]; then tail -5 $out/run.log; exit 0; fi
python - "$f" <<'PY' > $root/gpurun_out/l2_hit.txt
import collections, csv, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "TCC_REQ_sum":
        cnt[k] += 1
rows = sorted(acc.items(), key=lambda kv: -kv[1].get("TCC_REQ_sum", 0))[:14]
print("| kernel | dispatches | TCC requests / dispatch | hit rate |")
print("|---|---|---|---|")
for k, v in rows:
    hit, miss, req = v.get("TCC_HIT_sum", 0), v.get("TCC_MISS_sum", 0), v.get("TCC_REQ_sum", 0)
    n = max(cnt[k], 1)
    print("| `%s` | %d | %.3g | %.3f |" % (k[:60], n, req / n, hit / max(hit + miss, 1)))
PY
cat $root/gpurun_out/l2_hit.txt
rm -rf $out/p
