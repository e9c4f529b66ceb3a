mkdir -p gpurun_out
C="1,4096,4096,1"
for a in 0 9 0 9; do echo "== STREAM ABL=$a"; TG_ABL=$a TG_STREAM=1 TG_VARIANT=801 timeout 120 python tools/quick_bench.py --configs "$C" --iters 5 2>&1 | grep -E "stacked"; done
