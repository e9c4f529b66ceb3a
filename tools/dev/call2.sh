set -u
mkdir -p gpurun_out
bash tools/exp.sh "1,4096,4096,1,any4_rowwise,128 8,4096,4096,1,any4_rowwise,128 8,8192,8192,0,any4_rowwise,128 1,4096,4096,1,int4,128 1,4096,4096,1,mx4,32 16,4096,4096,1,any4_rowwise,128" full_base full_noslp 2>&1 | tee gpurun_out/ab_noslp.txt
ROUND=r03 bash tools/gpu_round.sh 2>&1 | tee gpurun_out/round.txt
