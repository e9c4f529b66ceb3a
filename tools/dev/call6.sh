set -u
mkdir -p gpurun_out
for f in tests/test_gpu_peer.py tests/test_gpu_fused.py tests/test_gpu_decode.py; do
  echo "== $f"; timeout 400 python -m pytest $f -x -q --timeout 120 2>&1 | tail -6
done
for v in "--interleave" "--interleave --bs 8" "--no-fuse --bs 8"; do
timeout 600 python tools/llama_decode_bench.py --config llama3_8b --steps 60 --warmup 10 $v 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['any4']['ms_per_token'])"
done
