#!/bin/bash
# GPU box: SQ counters of the split-bf16 stride-2 weight gradient next to the fp32 kernel (PMC passes only, no trace
# domains).  usage: scripts/split_bf16_pmc.sh <tag>
tag=${1:-rXX}
mode=${2:-s2only}
root=$(pwd)
out=$root/gpurun_out/split_pmc_$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
CMD="python $root/scripts/bench_split_bf16.py $mode"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $out/a -o a -- $CMD > $out/a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $out/b -o b -- $CMD > $out/b.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC --output-format csv -d $out/c -o c -- $CMD > $out/c.log 2>&1
cd $root
python scripts/pmc_summary.py $out/split_pmc.csv $(find $out -name "*counter_collection.csv")
python - $out/split_pmc.csv <<'PY'
import csv, sys
rows = [r for r in csv.reader(l for l in open(sys.argv[1]) if not l.startswith("#"))]
head = rows[0]
for r in rows[1:]:
    if "wgrad_s2" in r[0] or "conv_s2" in r[0] or "k_conv_mfma<2" in r[0] or "split_w" in r[0]:
        print(r[0], "dispatches", r[1])
        for n, v in zip(head[2:], r[2:]):
            if v:
                print("   %-28s %s" % (n, v))
PY
rm -rf $out/a $out/b $out/c
