set -u
timeout 1800 python -m pytest tests/test_gpu_fast.py -x -q -k "llama3_8b" 2>&1 | tail -5
