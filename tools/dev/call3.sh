set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fused.py -x -q 2>&1 | tail -15
timeout 600 python tools/llama_decode_bench.py --config llama3_8b --steps 60 --warmup 10 2>&1 | tail -2
timeout 600 python tools/llama_decode_bench.py --config llama3_8b --steps 60 --warmup 10 --interleave 2>&1 | tail -2
timeout 600 python tools/llama_decode_bench.py --config llama3_8b --steps 60 --warmup 10 --bs 8 --interleave 2>&1 | tail -2
timeout 600 python tools/llama_decode_bench.py --config llama3_8b --steps 60 --warmup 10 --bs 8 --no-fuse 2>&1 | tail -2
