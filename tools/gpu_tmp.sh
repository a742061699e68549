#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_style_prep.py tests/test_lres_models.py -m gpu -q -x --no-header -p no:cacheprovider 2>&1 | tail -2
LVG_BENCH_NO_TRAIN_LEGS=1 timeout 600 python bench.py --steps 30 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('value', d['value'], 'ms', d['ms_per_step'], 'roofline', d['roofline']['frac'])"
