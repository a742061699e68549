#!/bin/bash
# r06: sres leg of bench.py per filtered_lrelu routing (LVG_FLRELU_STRIP = 0 / 1 / ...), with the per-launch table.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for b in ${@:-0 1}; do
  LVG_FLRELU_STRIP=$b LVG_BENCH_VERBOSE=1 LVG_BENCH_LEGS=sres timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/r06_sres_ab_$b.log 2> gpurun_out/r06_sres_ab_$b.err
  echo "== STRIP=$b"; grep "^\[op\] filtered_lrelu" gpurun_out/r06_sres_ab_$b.err | cut -c1-140
  python - <<PY
import json
line=[l for l in open('gpurun_out/r06_sres_ab_$b.log') if l.startswith('{')][-1]
d=json.loads(line)['sres']
r=d['roofline']
print('STRIP=$b sres ms_per_step', d['ms_per_step'], 'flrelu frac', r['frac'], 'avg us', r['avg_launch_us'])
for k,v in r['families'].items(): print('   ', k, v)
PY
done 2>&1 | tee gpurun_out/r06_sres_ab.log
