# developer probe: sustained time + sclk + socket power per library variant:  bash dev/power_ab.sh "CFG" NAME NAME ...
set -u
mkdir -p gpurun_out
cfg=$1; shift
cp any4_amd/lib/libtinygemm_hip.so /tmp/orig.so
for v in "$@"; do
  if [ "$v" = orig ]; then cp /tmp/orig.so any4_amd/lib/libtinygemm_hip.so; else cp variants/$v.so any4_amd/lib/libtinygemm_hip.so; fi
  AB_HOLD=5 python tools/ab.py $cfg > /tmp/ab.out 2>&1 &
  pid=$!
  sleep 6.5
  s=""
  for i in 1 2 3 4; do
    s="$s $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | sed -E 's/.*\(([0-9]+)Mhz\).*/\1MHz/; s/.*Power \(W\): ([0-9.]+)/\1W/' | tr '\n' ' ')"
    sleep 0.4
  done
  wait $pid
  echo "=== $v : $(grep -v amdgpu.ids /tmp/ab.out | sed -E 's/.*  ([0-9.]+ us\/layer  [0-9.]+%).*/\1/') | $s"
done
cp /tmp/orig.so any4_amd/lib/libtinygemm_hip.so
