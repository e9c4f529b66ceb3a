timeout 300 python tools/quick_bench.py --configs "1,4096,4096,1;8,4096,4096,1;1,4096,4096,0" --qtype int8 --iters 3 2>&1 | grep -E "^m=|eager|stacked|steady"
