#!/bin/bash
# SQ wait-state counters of the 1-wave-per-SIMD kernels (one PMC pass, no trace domains).
# usage: scripts/pmc_sq.sh <tag>
set -u
tag=${1:-sq}
root=$(pwd)
out=$root/gpurun_out/pmc_$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES"
rocprofv3 --pmc $C --output-format csv -d $out/a -o a -- python $root/scripts/bench_convt.py child > $out/a.log 2>&1
SR_WINOGRAD=1 rocprofv3 --pmc $C --output-format csv -d $out/b -o b -- python $root/scripts/bench_wino.py child > $out/b.log 2>&1
SR_WINOGRAD=1 rocprofv3 --pmc $C --output-format csv -d $out/c -o c -- python $root/scripts/bench_wgrad_wino.py child > $out/c.log 2>&1
cd $root
python scripts/pmc_summary.py $out/sq.csv $out/a/a_counter_collection.csv $out/b/b_counter_collection.csv $out/c/c_counter_collection.csv
grep -E "^kernel|k_convt_fused|k_conv_wino|k_wgrad_wino,|k_conv_mfma<1; 2; 2" $out/sq.csv
