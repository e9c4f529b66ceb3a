set -u
timeout 900 python -m pytest tests/test_gpu_decode.py -x -q -k "accuracy_loop" 2>&1 | tail -5
timeout 600 python tools/eval_accuracy.py --arch tiny --quantize anyq --calibrate --seqlen 128 --n-tokens 16384 2>&1 | tail -2
timeout 600 python tools/eval_accuracy.py --arch tiny --quantize intq --seqlen 128 --n-tokens 16384 2>&1 | tail -1
timeout 600 python tools/eval_accuracy.py --arch tiny --quantize mx4 --seqlen 128 --n-tokens 16384 2>&1 | tail -1
timeout 900 python tools/hf_benchmark.py --arch llama3_8b --layers 4 --iters 20 --warmup 5 2>&1 | tail -12
