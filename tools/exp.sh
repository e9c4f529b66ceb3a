# scratch: one A/B visit to a GPU box (developer use).  Typical shape:
#   build variants with tools/dev/build_variant.sh NAME -DTG_DEV_MIN=<NSG> [-DTG_DEV_MR=4] [-DTG_DEV_LA=true] [-D...]
#   then, on the box, swap any4_amd/lib/libtinygemm_hip.so for each variants/NAME.so and run tools/dev/ab.py (oracle check +
#   steady timing); A/B numbers are only comparable within ONE visit (boxes differ by +-3 %).
#   usage: bash tools/exp.sh "CFG CFG ..." NAME NAME ...      (NAME "orig" = the installed library)
set -u
cfgs=$1; shift
cp any4_amd/lib/libtinygemm_hip.so /tmp/orig.so
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = orig ]; then cp /tmp/orig.so any4_amd/lib/libtinygemm_hip.so; else cp variants/$v.so any4_amd/lib/libtinygemm_hip.so; fi
  echo "=== $v (pass $rep)"; timeout 300 python tools/dev/ab.py $cfgs 2>&1 | grep -v amdgpu.ids
done
done
cp /tmp/orig.so any4_amd/lib/libtinygemm_hip.so
