set -u
mkdir -p gpurun_out
bash tools/exp.sh "8,4096,4096,1,any4_rowwise,128 4,4096,4096,1,any4_rowwise,128 2,4096,4096,1,any4_rowwise,128" m8_base m8_abl9 m8_abl10 m8_abl11 m8_abl5 m8_abl7 m8_abl1 m8_abl4 m8_abl6 m8_abl3 m8_alias 2>&1 | grep -v "^$" | tee gpurun_out/ab_m8_abl.txt
