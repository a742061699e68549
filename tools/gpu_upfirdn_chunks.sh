cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for ch in 0 1 2 3 4 6; do echo "== LVG_UPFIRDN_CHUNKS=$ch"; LVG_UPFIRDN_CHUNKS=$ch timeout 120 python tools/upfirdn_chunk_bench.py 2>&1 | grep "TB/s"; done | tee gpurun_out/r02_upfirdn_chunks.log
