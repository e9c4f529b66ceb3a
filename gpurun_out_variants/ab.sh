C="1,4096,4096,1;8,4096,4096,1;1,4096,4096,0"
cp any4_amd/lib/libtinygemm_hip.so /tmp/base.so
for v in base v5 base v5; do
  if [ $v = base ]; then cp /tmp/base.so any4_amd/lib/libtinygemm_hip.so; else cp gpurun_out_variants/lib_$v.so any4_amd/lib/libtinygemm_hip.so; fi
  echo "== $v: $(timeout 300 python tools/quick_bench.py --configs "$C" --iters 3 2>&1 | grep -E "steady" | awk '{print $2}' | tr '\n' ' ')"
done
cp gpurun_out_variants/lib_v5.so any4_amd/lib/libtinygemm_hip.so
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "stream or gemm_rm or m_sweep or identity" 2>&1 | tail -2
cp /tmp/base.so any4_amd/lib/libtinygemm_hip.so
