timeout 300 python tools/quick_bench.py --configs "1,4096,4096,1;1,4096,4096,0;8,4096,4096,0" --qtype int8 --iters 3 2>&1 | grep -E "^m=|steady"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k int8 2>&1 | tail -2
