cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_sres_w -o w -- python tools/sres_step.py 3 > gpurun_out/r02_sres_step.log 2>&1
f=$(find gpurun_out/prof_sres_w -name "*kernel_trace.csv" | head -1)
python tools/trace_window.py "$f" $(python -c "
import json
for l in open('gpurun_out/r02_sres_step.log'):
    if l.startswith('{'):
        d=json.loads(l); print(d['window_ms'], d['steps'])") > gpurun_out/r02_sres_window_stats.csv 2>&1
rm -rf gpurun_out/prof_sres_w
grep "^{" gpurun_out/r02_sres_step.log; head -40 gpurun_out/r02_sres_window_stats.csv | cut -c1-170
