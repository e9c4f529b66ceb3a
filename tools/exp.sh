# Scratch pad for one-off GPU experiments (run as: gpurun -- 'bash tools/exp.sh').  Kept in the tree because the profiles
# under profiles/ name it as their origin; the reproducible flow is tools/gpu_round.sh.  Example: A/B two builds of the
# library on ONE box (boxes differ by +-3 %, and the m = 1 kernel is sensitive to code generation):
#   cp any4_amd/lib/libtinygemm_hip.so /tmp/base.so
#   for v in base variant base variant; do cp /tmp/$v.so any4_amd/lib/libtinygemm_hip.so; python tools/quick_bench.py --configs "1,4096,4096,1" --iters 3 | grep steady; done
timeout 300 python tools/quick_bench.py --configs "1,4096,4096,1;8,4096,4096,1" --iters 3 2>&1 | grep -E "^m=|steady"
