set -u
cp any4_amd/lib/libtinygemm_hip.so /tmp/orig.so
for round in 1 2; do
for v in mr4n0 mr4n2 mr1n0; do
  cp variants/$v.so any4_amd/lib/libtinygemm_hip.so
  echo "=== $v $(timeout 300 python tools/quick_bench.py --configs "1,4096,4096,1" --L 256 --iters 5 2>&1 | grep -E "steady|==eager")"
done
done
cp /tmp/orig.so any4_amd/lib/libtinygemm_hip.so
