#!/bin/bash
# GPU box: kernel trace of the graph-replayed inversion step (BASELINE config[4]); per-step kernel table of the replays.
# usage: scripts/inversion_prof.sh <tag>
tag=${1:-rXX}
root=$(pwd)
out=$root/gpurun_out/inv_$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $out/trace -o inv -- python $root/scripts/inversion_replay_probe.py 40 > $out/run.log 2>&1
f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python $root/scripts/trace_summary.py $f 40 --after-gap 300 > $out/inversion_step_trace.txt 2>&1
rm -rf $out/trace
head -60 $out/inversion_step_trace.txt
tail -2 $out/run.log
