# like tools/exp.sh, with AB_HOLD seconds of load before the samples (sustained, power-capped state)
set -u
cfgs=$1; shift
cp any4_amd/lib/libtinygemm_hip.so /tmp/orig.so
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = orig ]; then cp /tmp/orig.so any4_amd/lib/libtinygemm_hip.so; else cp variants/$v.so any4_amd/lib/libtinygemm_hip.so; fi
  echo "=== $v (pass $rep)"; AB_HOLD=${AB_HOLD:-3} timeout 300 python tools/dev/ab.py $cfgs 2>&1 | grep -v amdgpu.ids
done
done
cp /tmp/orig.so any4_amd/lib/libtinygemm_hip.so
