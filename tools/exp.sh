# scratch: one A/B visit to a GPU box (developer use).  Typical shape:
#   build variants with tools/dev/build_variant.sh NAME -DTG_DEV_MIN=<NSG> [-DTG_DEV_MR=4] [-DTG_DEV_LA=true] [-D...]
#   then, on the box, swap any4_amd/lib/libtinygemm_hip.so for each variants/NAME.so and run tools/quick_bench.py;
#   A/B numbers are only comparable within ONE visit (boxes differ by +-3 %).
set -u
cp any4_amd/lib/libtinygemm_hip.so /tmp/orig.so
for v in "$@"; do
  cp variants/$v.so any4_amd/lib/libtinygemm_hip.so
  for cfg in "1,4096,4096,1" "8,4096,4096,1" "8,8192,8192,0"; do
    echo "=== $v $cfg $(timeout 300 python tools/quick_bench.py --configs "$cfg" --L 256 --iters 5 2>&1 | grep -E "steady|plan=" | tr '\n' ' ' | sed 's/on_right.*plan=/plan=/')"
  done
done
cp /tmp/orig.so any4_amd/lib/libtinygemm_hip.so
