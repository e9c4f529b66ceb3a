timeout 300 python tools/quick_bench.py --configs "1,4096,4096,1" --qtype mx4 --iters 3 2>&1 | grep -E "^m=|eager|stacked|steady"
timeout 300 python tools/quick_bench.py --configs "1,4096,4096,1;8,4096,4096,1" --iters 3 2>&1 | grep -E "^m=|eager|stacked|steady"
TG_STREAM=0 timeout 300 python tools/quick_bench.py --configs "1,4096,4096,1" --iters 3 2>&1 | grep -E "^m=|eager|stacked|steady"
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
