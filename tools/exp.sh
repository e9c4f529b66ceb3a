timeout 600 python -m pytest tests/test_gpu_decode.py -x -q 2>&1 | tail -6
timeout 900 python tools/llama_decode_bench.py --config llama3_8b --baseline 2>&1 | tail -1 | cut -c150-600
timeout 900 python tools/llama_decode_bench.py --config llama3_8b --start-pos 900 2>&1 | tail -1 | cut -c150-420
