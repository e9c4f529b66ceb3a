timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
C="1,6144,4096,1;1,4096,4096,1;1,28672,4096,1;1,4096,14336,1;1,8192,8192,1"
echo "== sk<=16"; timeout 300 python tools/quick_bench.py --configs "$C" --iters 3 --settle 0 --L 32 2>&1 | grep -E "graph" | awk '{print $2}' | tr '\n' ' '; echo
echo "== sk=8";  TG_SK=8 timeout 300 python tools/quick_bench.py --configs "$C" --iters 3 --settle 0 --L 32 2>&1 | grep -E "graph" | awk '{print $2}' | tr '\n' ' '; echo
timeout 900 python tools/llama_decode_bench.py --config llama3_8b 2>&1 | tail -1 | cut -c260-420
