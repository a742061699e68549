# Round-end evidence in one gpurun call: the GPU suite, smoke, the default bench line as the driver runs it (tools/gpu_full_suite.sh), one
# rocprofv3 --kernel-trace --stats table per bench leg (tools/gpu_leg_profiles.sh), and the HBM traffic passes (tools/gpu_traffic.sh).
export ROUND=${ROUND:-r05}
bash tools/gpu_full_suite.sh
bash tools/gpu_leg_profiles.sh
bash tools/gpu_traffic.sh > gpurun_out/${ROUND}_traffic_run.log 2>&1
tail -3 gpurun_out/${ROUND}_traffic_run.log | cut -c1-300
