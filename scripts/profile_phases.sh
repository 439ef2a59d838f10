#!/bin/bash
# GPU box: kernel trace of each captured phase of the config[2] iteration (8 replays each), full kernel tables.
# usage: scripts/profile_phases.sh <tag> [phases...]
tag=${1:-rXX}; shift
phases=${@:-d g path r1}
root=$(pwd)
out=$root/gpurun_out/phases_$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
for p in $phases; do
  rocprofv3 --kernel-trace --output-format csv -d $out/$p -o $p -- python $root/scripts/phase_trace.py $p 8 > $out/$p.log 2>&1
  f=$(find $out/$p -name "*kernel_trace.csv" | head -1)
  python $root/scripts/trace_summary.py $f 8 --after-gap 400 > $out/${p}_phase_trace.txt 2>&1
  rm -rf $out/$p
done
