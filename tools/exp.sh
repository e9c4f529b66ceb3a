timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "stream or gemm_m_sweep or full_size" 2>&1 | tail -3
C="8,4096,4096,1;16,2048,2048,1;4,8192,8192,1;8,4096,4096,0"
timeout 300 python tools/quick_bench.py --configs "$C" --iters 3 --L 128 2>&1 | grep -E "^m=|steady"
