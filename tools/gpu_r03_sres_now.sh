#!/bin/bash
# r03: the sres generator update now: frames/s (hipGraph replay), then the per-kernel table of the timed window under rocprofv3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tag=${1:-now}
timeout 300 python tools/sres_step.py 6 2>&1 | grep "^{" | tee gpurun_out/r03_sres_step_$tag.log
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_sres_w -o w -- python tools/sres_step.py 3 > gpurun_out/r03_sres_step_prof.log 2>&1
f=$(find gpurun_out/prof_sres_w -name "*kernel_trace.csv" | head -1)
python tools/trace_window.py "$f" $(python -c "
import json
for l in open('gpurun_out/r03_sres_step_prof.log'):
    if l.startswith('{'):
        d=json.loads(l); print(d['window_ms'], d['steps'])") > gpurun_out/r03_sres_window_stats_$tag.csv 2>&1
rm -rf gpurun_out/prof_sres_w
grep "^{" gpurun_out/r03_sres_step_prof.log; head -45 gpurun_out/r03_sres_window_stats_$tag.csv | cut -c1-200
