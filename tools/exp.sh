C="1,4096,4096,1"
echo "== m=1 privx"; timeout 300 python tools/quick_bench.py --configs "$C" --iters 3 2>&1 | grep -E "steady"
echo "== m=1 XRES 16-wave WG"; TG_XRES=2 timeout 300 python tools/quick_bench.py --configs "$C" --iters 3 2>&1 | grep -E "steady"
