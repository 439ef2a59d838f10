#!/bin/bash
# GPU box: instruction-mix / busy counters of the rasterizer kernels (PMC passes only, no trace domains) and the HBM
# traffic passes.  usage: scripts/raster_pmc.sh <tag>
tag=${1:-rXX}
root=$(pwd)
out=$root/gpurun_out/raster_pmc_$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
CMD="python $root/scripts/raster_bench.py 6"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $out/a -o a -- $CMD > $out/a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS --output-format csv -d $out/b -o b -- $CMD > $out/b.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/c -o c -- $CMD > $out/c.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/d -o d -- $CMD > $out/d.log 2>&1
rocprofv3 --pmc TA_BUSY_avr TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $out/e -o e -- $CMD > $out/e.log 2>&1
cd $root
python scripts/pmc_summary.py $out/raster_pmc.csv $(find $out -name "*counter_collection.csv")
grep -E "^kernel|k_tile|k_grad|k_first" $out/raster_pmc.csv
rm -rf $out/a $out/b $out/c $out/d $out/e
