C="1,4096,4096,1;8,4096,4096,1;2,4096,4096,1;1,4096,4096,0;1,8192,8192,1;8,8192,8192,0"
timeout 300 python tools/quick_bench.py --configs "$C" --iters 3 2>&1 | grep -E "^m=|steady"
