cd /root/repo
one() { python bench.py --steps 12 --warmup 4 --no-train --no-raster --no-inversion --no-pmc --no-split-bf16 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
SR_BLUR_BWD_FUSED=0 one two-kernel
one fused
SR_BLUR_BWD_FUSED=0 one two-kernel
one fused
SR_BLUR_BWD_FUSED=0 SR_FIR_ALIGNED=0 one two-kernel-unaligned
cd /tmp && export TMPDIR=/tmp
for m in 0 1; do
SR_BLUR_BWD_FUSED=$m rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/firprof$m -o p -- python /root/repo/bench.py --steps 6 --warmup 2 --no-train --no-raster --no-inversion --no-pmc --no-split-bf16 > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob('/root/repo/gpurun_out/firprof$m/**/*kernel_stats.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('mode $m total kernel ms/step', tot/8/1e6)
for r in rows:
    n=r['Name']
    if any(k in n for k in ('fir4','nba_bwd','nba_finish','rowdot')):
        print('  %-60s calls %5s  total/step %.3f ms  avg %.1f us' % (n[:60], r['Calls'], float(r['TotalDurationNs'])/8/1e6, float(r['AverageNs'])/1e3))
PY
done
