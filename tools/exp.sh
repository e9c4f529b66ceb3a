for d in 4096 8192; do for q in anyq intq; do echo "== $d $q"; timeout 300 python tools/microbenchmark.py --input-dim $d --output-dim $d --quantize $q 2>&1 | grep -v amdgpu | tail -5; done; done
echo "== nf4-style global LUT"; timeout 300 python tools/microbenchmark.py --quantize anyq --quantize-args per_row=False 2>&1 | tail -3
