set -u
mkdir -p gpurun_out
bash tools/exp.sh "8,8192,8192,0,any4_rowwise,128 1,8192,8192,0,any4_rowwise,128 8,4096,4096,0,any4_rowwise,128" la_base la_abl9 la_abl7 la_abl1 la_abl4 la_abl6 la_abl3 la_abl8 2>&1 | grep -v "^$" | tee gpurun_out/ab_la_abl.txt
