#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x --no-header -p no:cacheprovider 2>&1 | tail -8 | tee gpurun_out/r03_gpu_tests_mid.log
timeout 900 python bench.py > gpurun_out/r03_bench_mid.log 2> gpurun_out/r03_bench_mid.err; tail -c 600 gpurun_out/r03_bench_mid.err; python - <<'P'
import json
for l in open('gpurun_out/r03_bench_mid.log'):
    if l.startswith('{'):
        d=json.loads(l)
        print('value',d['value'],'ms',d['ms_per_step'],'roofline',d['roofline']['frac'])
        print('roofline_hbm',json.dumps(d.get('roofline_hbm'))[:300])
        for k in ('sres','train_sres','train_lres','forward_only'):
            if k in d: print(k, d[k].get('value'), d[k].get('ms_per_step'), json.dumps(d[k].get('roofline',{}))[:200])
P
