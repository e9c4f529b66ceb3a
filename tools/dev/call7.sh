set -u
mkdir -p gpurun_out
for f in tests/test_gpu_peer.py; do echo "== $f"; timeout 400 python -m pytest $f -x -q --timeout 120 2>&1 | tail -4; done
echo "== test_gpu_fast (9..16 rows, bench shapes)"; timeout 600 python -m pytest tests/test_gpu_fast.py -x -q --timeout 120 -k "9_to_16 or many_rows or benchmarked_launch_shape" 2>&1 | tail -6
bash tools/exp.sh "16,4096,4096,1,any4_rowwise,128 12,4096,4096,1,any4_rowwise,128 9,4096,4096,1,any4_rowwise,128 16,8192,8192,1,any4_rowwise,128,64" b16_r2c4 b16_r2c1 b16_r2c8 b16_r3c4 b16_r4c4 2>&1 | tee gpurun_out/ab_b16.txt
