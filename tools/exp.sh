set -u
timeout 1800 python -m pytest tests/test_gpu_fast.py -x -q 2>&1 | tail -3
cp any4_amd/lib/libtinygemm_hip.so /tmp/orig.so
for v in ew0 ew1 ew0 ew1; do
  cp variants/$v.so any4_amd/lib/libtinygemm_hip.so
  for cfg in "2,4096,4096,1" "8,4096,4096,1" "8,8192,8192,1"; do
  echo "=== $v $cfg $(timeout 300 python tools/quick_bench.py --configs "$cfg" --L 256 --iters 5 2>&1 | grep -E "steady|plan=" | tr '\n' ' ' | sed 's/on_right.*plan=/plan=/')"
  done
done
cp /tmp/orig.so any4_amd/lib/libtinygemm_hip.so
