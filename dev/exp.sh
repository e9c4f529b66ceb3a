# One same-box A/B visit: every variant library (dev/build_variant.sh NAME ... -> variants/NAME.so; "orig" = the installed one) runs
# tools/ab.py on the given shapes, twice round.  Numbers are only comparable within ONE visit (boxes differ by +-3 %), and only
# in the sustained state: AB_HOLD seconds of load in front of the samples (default 3; 0 = the transient after idle).
#   usage: bash dev/exp.sh "CFG CFG ..." NAME NAME ...        CFG = m,n,k,on_right,qtype,g[,layers]
set -u
cfgs=$1; shift
cp any4_amd/lib/libtinygemm_hip.so /tmp/orig.so
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = orig ]; then cp /tmp/orig.so any4_amd/lib/libtinygemm_hip.so; else cp variants/$v.so any4_amd/lib/libtinygemm_hip.so; fi
  echo "=== $v (pass $rep)"; AB_HOLD=${AB_HOLD:-3} timeout 300 python tools/ab.py $cfgs 2>&1 | grep -v amdgpu.ids
done
done
cp /tmp/orig.so any4_amd/lib/libtinygemm_hip.so
