set -u
timeout 1800 python -m pytest tests/test_gpu_fast.py tests/test_gpu_decode.py -x -q 2>&1 | tail -3
echo "== decode fast"; timeout 600 python tools/llama_decode_bench.py --config llama3_8b --steps 50 --warmup 10 2>&1 | tail -3
echo "== decode reference"; ANY4_NUMERICS=reference timeout 600 python tools/llama_decode_bench.py --config llama3_8b --steps 50 --warmup 10 2>&1 | tail -3
echo "== decode fast bs8"; timeout 600 python tools/llama_decode_bench.py --config llama3_8b --bs 8 --steps 50 --warmup 10 2>&1 | tail -2
echo "== decode reference bs8"; ANY4_NUMERICS=reference timeout 600 python tools/llama_decode_bench.py --config llama3_8b --bs 8 --steps 50 --warmup 10 2>&1 | tail -2
