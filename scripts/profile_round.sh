#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + the separate PMC passes that bench.py's roofline blocks cite
# (generator leg + rasterizer leg in one command), and a kernel trace of the hipGraph-replayed G+D iteration.
# usage: scripts/profile_round.sh <tag>   (outputs under gpurun_out/prof_<tag>)
set -u
tag=${1:-rXX}
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
BENCH="python $root/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-train --no-inversion --no-pmc --no-split-bf16"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o $tag -- $BENCH > $out/trace.log 2>&1
# PMC passes: never combined with the trace domains, one TCC counter per pass
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -o $tag -- $BENCH > $out/pmc_$c.log 2>&1
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_mfma -o $tag -- $BENCH > $out/pmc_mfma.log 2>&1
# the full G+D iteration (BASELINE config[2], 4 images, graph replay): 2 warm-up + 16 timed iterations
rocprofv3 --kernel-trace --stats --output-format csv -d $out/train -o $tag -- python $root/scripts/train_step_probe.py 16 4 > $out/train.log 2>&1
cd $root
python scripts/trace_summary.py $out/train/${tag}_kernel_trace.csv 16 0.45 40 > $out/train_summary.txt 2>&1
find $out -name '*.csv' | head -20
grep -h '"metric"' $out/*.log | cut -c1-200
tail -1 $out/train.log
