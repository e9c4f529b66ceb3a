set -u
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout 900 python bench.py > gpurun_out/bench_try.json 2> gpurun_out/bench_try.err; tail -3 gpurun_out/bench_try.err
