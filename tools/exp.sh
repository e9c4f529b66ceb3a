C="1,4096,4096,1;8,4096,4096,1;1,4096,4096,0"
echo "== new"; timeout 300 python tools/quick_bench.py --configs "$C" --iters 3 2>&1 | grep -E "steady"
cp any4_amd/lib/libtinygemm_hip.so /tmp/new.so; cp gpurun_out_lib_old.so any4_amd/lib/libtinygemm_hip.so
echo "== old"; timeout 300 python tools/quick_bench.py --configs "$C" --iters 3 2>&1 | grep -E "steady"
cp /tmp/new.so any4_amd/lib/libtinygemm_hip.so
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
