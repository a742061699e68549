cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -f gpurun_out/r05_thin_ab.log
for v in 1 0 1 0; do
  LVG_THIN_POINTWISE=$v LVG_BENCH_LEGS=train_lres timeout 400 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('LVG_THIN_POINTWISE=$v', 'step', d['ms_per_step'], 'ms', d['value'], 'frames/s; train_lres', d['train_lres'].get('ms_per_step'), 'ms', d['train_lres'].get('error', ''))
" | tee -a gpurun_out/r05_thin_ab.log
done
