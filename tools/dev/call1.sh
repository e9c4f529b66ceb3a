# GPU visit: smoke, gpu tests, bench line, m=8 A/B variants, LDS out-of-range probe
set -u
mkdir -p gpurun_out
python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke.log
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/gpu_tests.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?"; tail -3 gpurun_out/bench.err; wc -c gpurun_out/bench.log
./variants/lds_oob > gpurun_out/lds_oob.txt 2>&1; cat gpurun_out/lds_oob.txt
bash tools/exp.sh "8,4096,4096,1,any4_rowwise,128 8,8192,8192,1,any4_rowwise,128" m8_base m8_xcd m8_xcd_c8 m8_noslp m8_xcd_noslp 2>&1 | tee gpurun_out/ab_m8.txt
