timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
C="1,6144,4096,1;1,4096,4096,1;1,28672,4096,1;1,4096,14336,1;1,8192,8192,1;1,4096,4096,0;1,6144,4096,0;1,28672,4096,0;1,4096,14336,0"
echo "== default"; timeout 300 python tools/quick_bench.py --configs "$C" --iters 3 --settle 0 --L 32 2>&1 | grep -E "^m=|graph"
timeout 900 python tools/llama_decode_bench.py --config llama3_8b 2>&1 | tail -1
timeout 900 python tools/llama_decode_bench.py --config llama3_8b --kernel linear_y_f16RM_W_any4TC_x_f16RM 2>&1 | tail -1
