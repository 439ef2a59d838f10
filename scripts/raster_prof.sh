#!/bin/bash
# GPU box: rocprofv3 kernel stats of the rasterizer leg alone. usage: scripts/raster_prof.sh <tag>
tag=${1:-rXX}
root=$(pwd)
out=$root/gpurun_out/raster_$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o $tag -- python $root/scripts/raster_bench.py 20 > $out/run.log 2>&1
f=$(find $out/trace -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY' > $out/kernel_stats.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-70s calls %6s  avg_us %9.2f  total_ms %8.3f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
cp $f $out/${tag}_raster_kernel_stats.csv
rm -rf $out/trace
cat $out/kernel_stats.txt
tail -1 $out/run.log
