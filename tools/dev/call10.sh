set -u
mkdir -p gpurun_out
bash tools/exp.sh "8,4096,4096,1,any4_rowwise,128 4,4096,4096,1,any4_rowwise,128 2,4096,4096,1,any4_rowwise,128" m8_base w16_m8 2>&1 | grep -v "^$" | tee gpurun_out/ab_w16.txt
bash tools/exp.sh "1,4096,4096,1,any4_rowwise,128" base_m1 w16_m1 2>&1 | grep -v "^$" | tee -a gpurun_out/ab_w16.txt
