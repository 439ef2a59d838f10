#!/bin/bash
# GPU box: where is the GPU idle inside the replayed config[2] iteration / the inversion step?  (scripts/trace_gaps_all.py)
export TMPDIR=/tmp
root=$(pwd); out=$root/gpurun_out/gaps; mkdir -p $out; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $out/train -o t -- python $root/scripts/train_step_probe.py 8 4 > $out/train.log 2>&1
f=$(find $out/train -name "*kernel_trace.csv" | head -1)
python $root/scripts/trace_gaps_all.py $f 0.6 15 > $out/train_gaps.txt 2>&1
rm -rf $out/train
