#!/bin/bash
# GPU box: kernel durations of the rasterizer gradient at a given batch under the three gather modes
# usage: scripts/raster_prof_b.sh <batch>
bs=${1:-1}
root=$(pwd)
export TMPDIR=/tmp
cd /tmp
for m in "SR_RASTER_GRAD_SLOTS=0" "SR_RASTER_GRAD_EAGER=0" "SR_RASTER_GRAD_EAGER=1"; do
  out=/tmp/rp_$$; rm -rf $out
  env $m rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python $root/scripts/raster_bench.py 40 $bs > /dev/null 2>&1
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  echo "== batch $bs $m"
  python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r["Name"] for k in ("k_grad_pix", "k_grad_vert", "k_grad_big", "k_first_pix", "k_slot")):
        print("   %-44s calls %4s avg_us %8.2f" % (r["Name"].split("(")[0][-44:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  rm -rf $out
done
