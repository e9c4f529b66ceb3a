timeout 300 python tools/quick_bench.py --configs "8,4096,4096,1;16,4096,4096,1;8,8192,8192,0" --iters 5 2>&1 | grep -E "^m=|stacked"
