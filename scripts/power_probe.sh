#!/bin/bash
# GPU box: shader clock and socket power sampled while the headline step runs (is the step power-managed?).
# usage: bash scripts/power_probe.sh [extra bench flags]   -> gpurun_out/power_probe.txt
cd /root/repo
out=gpurun_out/power_probe.txt
mkdir -p gpurun_out
rocm-smi --showpower --showclocks --showmaxpower 2>/dev/null | grep -v "^=\|^$" | head -20 > $out
( for i in $(seq 1 400); do
    echo "t=$(date +%s.%N) $(rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'sclk|Power' | tr -s ' ' | tr '\n' '|')"
    sleep 0.05
  done ) > gpurun_out/power_samples.txt &
SAMPLER=$!
python bench.py --steps 300 --warmup 5 --no-train --no-raster --no-inversion --no-pmc --no-split-bf16 "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'])" >> $out
kill $SAMPLER 2>/dev/null
wait $SAMPLER 2>/dev/null
wc -l gpurun_out/power_samples.txt >> $out
python - <<'PY' >> gpurun_out/power_probe.txt
import re
rows=[]
for l in open('/root/repo/gpurun_out/power_samples.txt'):
    m=re.search(r'sclk clock level: \d+: \((\d+)Mhz\)', l)
    p=re.search(r'Power \(W\): ([\d.]+)', l)
    if m and p: rows.append((float(l.split()[0][2:]), int(m.group(1)), float(p.group(1))))
if rows:
    t0=rows[0][0]
    import statistics
    busy=[r for r in rows if r[2] > 0.5*max(x[2] for x in rows)]
    print('samples', len(rows), 'busy', len(busy))
    print('sclk MHz busy: min %d median %d max %d' % (min(r[1] for r in busy), statistics.median(r[1] for r in busy), max(r[1] for r in busy)))
    print('power W busy: min %.0f median %.0f max %.0f' % (min(r[2] for r in busy), statistics.median(r[2] for r in busy), max(r[2] for r in busy)))
    print('idle sclk/power:', [(r[1], r[2]) for r in rows[:3]])
else:
    print('no parsable samples; first lines:')
    print(open('/root/repo/gpurun_out/power_samples.txt').read()[:600])
PY
cat $out
