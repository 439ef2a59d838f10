#!/bin/bash
# GPU box: kernel SEQUENCE (scripts/trace_sequence.py) of one replay of the inversion step and of the named config[2] phases.
# usage: scripts/sequence_prof.sh <tag> [phases...]
tag=${1:-rXX}; shift
phases=${@:-g}
root=$(pwd)
out=$root/gpurun_out/seq_$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $out/inv -o inv -- python $root/scripts/inversion_replay_probe.py 8 > $out/inv.log 2>&1
f=$(find $out/inv -name "*kernel_trace.csv" | head -1)
python $root/scripts/trace_sequence.py $f 8 > $out/inversion_sequence.txt 2>&1
python $root/scripts/trace_summary.py $f 8 --after-gap 300 > $out/inversion_step_trace.txt 2>&1
rm -rf $out/inv
for p in $phases; do
  rocprofv3 --kernel-trace --output-format csv -d $out/$p -o $p -- python $root/scripts/phase_trace.py $p 4 > $out/$p.log 2>&1
  f=$(find $out/$p -name "*kernel_trace.csv" | head -1)
  python $root/scripts/trace_sequence.py $f 4 > $out/${p}_sequence.txt 2>&1
  rm -rf $out/$p
done
