cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_noise_bank.py tests/test_lres_models.py -m gpu -q --no-header -x 2>&1 | tail -8 > gpurun_out/r05_noise_tests.log; cat gpurun_out/r05_noise_tests.log
timeout 300 python tools/noise_bank_time.py > gpurun_out/r05_noise_bank_time.log 2>&1; cat gpurun_out/r05_noise_bank_time.log
for v in 1 0 1; do
  LVG_NOISE_BANK=$v LVG_BENCH_LEGS=forward_only timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('LVG_NOISE_BANK=$v', 'step', d['ms_per_step'], 'ms', d['value'], 'frames/s; forward_only', d['forward_only']['ms_per_step'], 'ms', d['forward_only']['value'])
" | tee -a gpurun_out/r05_noise_bank_ab.log
done
