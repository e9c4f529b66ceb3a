set -u
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for extra in "--qtype mx4" "--qtype int4 --g 32" "--g 64" "--g 256" "--g 32" "--qtype any4_global --g 64"; do
for cfg in "1,4096,4096,1" "8,4096,4096,1"; do
  timeout 300 python tools/quick_bench.py --configs "$cfg" $extra --L 256 --iters 5 2>&1 | grep -E "plan=|steady" | tr '\n' ' ' | sed 's/on_right=1//;s/stacked==eager:False//'; echo
done
done
