# developer probe: GPU clock / power while a steady stacked launch runs (rocm-smi sampled every 0.5 s)
set -u
mkdir -p gpurun_out
for cfg in "1,4096,4096,1,any4_rowwise,128" "8,4096,4096,1,any4_rowwise,128" "8,8192,8192,0,any4_rowwise,128"; do
  echo "=== $cfg"
  AB_HOLD=8 python tools/ab.py $cfg > /tmp/ab.out 2>&1 &
  pid=$!
  sleep 6
  for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
    rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (junction|memory)" | tr -s ' ' | tr '\n' ';'; echo
    sleep 0.5
  done
  wait $pid
  grep -v amdgpu.ids /tmp/ab.out
done
