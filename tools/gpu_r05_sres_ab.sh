#!/bin/bash
# r05: sres leg of bench.py with the row-band filtered_lrelu kernel off (0) / default routing (1) / everything it can take (2).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for b in ${@:-0 1 2}; do
  LVG_FLRELU_BAND=$b LVG_BENCH_LEGS=sres timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/r05_sres_ab_$b.log 2> gpurun_out/r05_sres_ab_$b.err
  python - <<PY
import json
line=[l for l in open('gpurun_out/r05_sres_ab_$b.log') if l.startswith('{')][-1]
d=json.loads(line)['sres']
r=d['roofline']
print('BAND=$b sres ms_per_step', d['ms_per_step'], 'flrelu frac', r['frac'], 'avg us', r['avg_launch_us'])
for k,v in r['families'].items(): print('   ', k, v)
PY
done 2>&1 | tee gpurun_out/r05_sres_ab.log
