set -u
mkdir -p gpurun_out
show() { python - "$1" <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
s=d['single_layer_launch']
print(sys.argv[1],'value',d['value'],'single b2b',s['us_per_launch_back_to_back'],'cold',s['us_cold_event_pair'],'graph',s['us_per_launch_in_hipgraph'],'floor',s['us_launch_floor_back_to_back'],'a-side graph',s['int4_a_side']['us_per_launch_in_hipgraph'],'decode',d['decode_llama3_8b']['ms_per_token'], 'm8', d['m8']['frac'])
P
}
python bench.py --no-pmc --no-cpu-baseline --steps 50 --warmup 10 > gpurun_out/b_def.json 2>/dev/null; show gpurun_out/b_def.json
HIP_FORCE_DEV_KERNARG=1 python bench.py --no-pmc --no-cpu-baseline --steps 50 --warmup 10 > gpurun_out/b_devk1.json 2>/dev/null; show gpurun_out/b_devk1.json
HIP_FORCE_DEV_KERNARG=0 python bench.py --no-pmc --no-cpu-baseline --steps 50 --warmup 10 > gpurun_out/b_devk0.json 2>/dev/null; show gpurun_out/b_devk0.json
