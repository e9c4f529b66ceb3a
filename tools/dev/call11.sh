set -u
mkdir -p gpurun_out
for i in 1 2 3 4 5 6; do python tools/dev/probe_mode.py 2>&1 | grep "^m="; done | tee gpurun_out/probe_mode.txt
for p in 4096 65536 1048576 3145728; do PROBE_PAD=$p python tools/dev/probe_mode.py 2>&1 | grep "^m="; done | tee -a gpurun_out/probe_mode.txt
bash tools/exp.sh "16,4096,4096,1,any4_rowwise,128 12,4096,4096,1,any4_rowwise,128 8,4096,4096,1,any4_rowwise,128,512 16,8192,8192,1,any4_rowwise,128,64" b16_r2c4 b16_r2c1 b16_r2c8 b16_r3c4 b16_r4c4 b16_r3c1 2>&1 | grep -v "^$" | tee gpurun_out/ab_b16.txt
