#!/bin/bash
# r03: per-kernel times of the ADA pipeline benchmark (fused forward, gather-form adjoint, the composition's kernels for comparison)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ada -o a -- python tools/ada_bench.py > /dev/null 2>&1
python - <<'P' | tee gpurun_out/r03_ada_kernel_stats.log
import csv, glob
f = glob.glob('/tmp/prof_ada/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print('calls, total ms, average us, share, kernel   (tools/ada_bench.py: fused x2, composed, hybrid; 16 x 24 planes of 144 x 256)')
for r in rows[:16]:
    print(f"{r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['AverageNs'])/1e3:9.1f} us {r['Percentage']:>6s}%  {r['Name'][:100]}")
P
