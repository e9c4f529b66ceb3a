set -u
mkdir -p gpurun_out
python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke.log
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/gpu_tests.log
timeout 900 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?"; tail -3 gpurun_out/bench.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/bench.log').read().strip().splitlines()[-1])
print('value',d['value'],'frac',d['roofline']['frac'])
for k in ('m8','m16','config3','int4','nf4','mx4'):
    print(k,d[k]['us_per_layer'],d[k]['frac'],d[k]['kernel_plan'])
print('ref',d['reference_numerics']['m1']['frac'],d['reference_numerics']['m8']['frac'])
print('single',d['single_layer_launch']['us_per_launch_back_to_back'],d['single_layer_launch']['us_per_launch_in_hipgraph'])
print('decode',d['decode_llama3_8b']['ms_per_token'])
P
