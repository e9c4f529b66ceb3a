set -u
timeout 1800 python -m pytest tests/test_gpu_fast.py -x -q 2>&1 | tail -3
for cfg in "1,4096,4096,1" "8,4096,4096,1" "8,8192,8192,1" "8,8192,8192,0" "1,8192,8192,0" "16,8192,8192,0" "8,4096,4096,0"; do
  timeout 300 python tools/quick_bench.py --configs "$cfg" --L 256 --iters 5 2>&1 | grep -E "plan=|steady" | tr '\n' ' ' | sed 's/stacked==eager:False//'; echo
done
for extra in "--g 64" "--qtype mx4" "--qtype int4 --g 32"; do
for cfg in "1,4096,4096,1" "8,4096,4096,1"; do
  timeout 300 python tools/quick_bench.py --configs $cfg $extra --L 256 --iters 5 2>&1 | grep -E "plan=|steady" | tr '\n' ' ' | sed 's/stacked==eager:False//'; echo
done
done
