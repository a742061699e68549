cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/traffic
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/traffic/sres_f -o p -- python tools/sres_step.py 1 > gpurun_out/traffic/sres_f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/traffic/sres_w -o p -- python tools/sres_step.py 1 > gpurun_out/traffic/sres_w.log 2>&1
python tools/pmc_traffic.py $(find gpurun_out/traffic/sres_f -name "*counter_collection.csv") $(find gpurun_out/traffic/sres_w -name "*counter_collection.csv") gpurun_out/r06_traffic_sres.json > /dev/null
rm -rf gpurun_out/traffic/*_f gpurun_out/traffic/*_w
python -c "
import json; d=json.load(open('gpurun_out/r06_traffic_sres.json'))
for k in ['filtered_lrelu_strip','filtered_lrelu_band','filtered_lrelu_wave','filtered_lrelu_fused16']: print(k, d.get(k), d.get(k+'_detail'))"
