# Scratch pad for one-off GPU experiments (run as: gpurun -- 'bash tools/exp.sh').
set -u
mkdir -p gpurun_out
python __graft_entry__.py smoke 2>&1 | tail -3
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
timeout 300 python tools/quick_bench.py --configs "1,4096,4096,1" --L 256 --iters 5 2>&1 | grep -E "^m=|steady|eager"
