timeout 600 python -m pytest tests/test_gpu_decode.py -x -q -k hf_llama 2>&1 | tail -5
timeout 600 python tools/microbenchmark.py --quantize anyq 2>&1 | grep -v amdgpu | tail -6
timeout 600 python tools/microbenchmark.py --quantize intq 2>&1 | tail -5
timeout 600 python tools/microbenchmark.py --quantize int8 2>&1 | tail -5
timeout 1500 python tools/hf_benchmark.py --arch llama3_8b --layers 8 2>&1 | tail -9
