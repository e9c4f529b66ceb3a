set -u
timeout 1800 python -m pytest tests/test_gpu_fast.py -x -q 2>&1 | tail -3
for extra in "--g 64" "--g 64 --inner 2" "--g 32 --inner 2" "--g 128 --inner 2" "--g 256 --inner 8" "--g 128 --inner 8"; do
for cfg in "1,4096,4096,1" "8,4096,4096,1"; do
  timeout 300 python tools/quick_bench.py --configs $cfg $extra --L 256 --iters 5 2>&1 | grep -E "plan=|steady" | tr '\n' ' ' | sed 's/stacked==eager:False//'; echo
done
done
