mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 300 python tools/quick_bench.py --configs "1,4096,4096,1;2,4096,4096,1;4,4096,4096,1;8,4096,4096,1;16,4096,4096,1;1,4096,4096,0;8,8192,8192,0;1,8192,8192,1" --iters 5 2>&1 | grep -E "^m=|stacked"
