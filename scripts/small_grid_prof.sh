#!/bin/bash
# GPU box: launches with a grid smaller than the chip that still take long (scripts/small_grid_census.py), for the inversion
# replay and the named config[2] phases.   usage: scripts/small_grid_prof.sh <tag> [phases...]
tag=${1:-rXX}; shift
phases=${@:-g path d}
root=$(pwd)
out=$root/gpurun_out/sg_$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $out/inv -o inv -- python $root/scripts/inversion_replay_probe.py 40 > /dev/null 2>&1
python $root/scripts/small_grid_census.py $(find $out/inv -name "*kernel_trace.csv" | head -1) 256 8 > $out/inversion.txt 2>&1
rm -rf $out/inv
for p in $phases; do
  rocprofv3 --kernel-trace --output-format csv -d $out/$p -o $p -- python $root/scripts/phase_trace.py $p 8 > /dev/null 2>&1
  python $root/scripts/small_grid_census.py $(find $out/$p -name "*kernel_trace.csv" | head -1) 256 8 > $out/$p.txt 2>&1
  rm -rf $out/$p
done
