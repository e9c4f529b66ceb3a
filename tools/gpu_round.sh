# One GPU-box visit: smoke, gpu tests, bench line, rocprofv3 kernel stats of the bench run and of every leg, PMC counters
# (MfmaUtil, VALU / LDS / L2, HBM bytes) of the bench kernel and of the m = 8 / config 3 / m = 16 legs.  Writes gpurun_out/.
set -u
RN=${ROUND:-r06}
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd $R
python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke.log
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/gpu_tests.log
timeout 900 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?"; tail -3 gpurun_out/bench.err; cp gpurun_out/bench.log gpurun_out/${RN}_bench_line.json
cd /tmp && export TMPDIR=/tmp
# kernel stats: the roofline kernel alone, then the whole default run (every leg's kernels: pair16, stream, config 4, decode glue)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/stats -o bench -- python $R/bench.py --roofline-only > $R/gpurun_out/prof_stats.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/legs -o bench -- python $R/bench.py --no-pmc --no-cpu-baseline > $R/gpurun_out/prof_legs.log 2>&1
cd $R
cp gpurun_out/prof/stats/*/bench_kernel_stats.csv gpurun_out/${RN}_bench_kernel_stats.csv 2>/dev/null || cp gpurun_out/prof/stats/bench_kernel_stats.csv gpurun_out/${RN}_bench_kernel_stats.csv
cp gpurun_out/prof/legs/*/bench_kernel_stats.csv gpurun_out/${RN}_legs_kernel_stats.csv 2>/dev/null || cp gpurun_out/prof/legs/bench_kernel_stats.csv gpurun_out/${RN}_legs_kernel_stats.csv
head -4 gpurun_out/${RN}_bench_kernel_stats.csv | cut -c1-250
# counters
python tools/collect_counters.py --out gpurun_out/${RN}_counters_bench_m1.json --match w4_gemm_pair_kernel --label "bench kernel: m=1 Bint4 4096^2, 512 layers per launch" -- python bench.py --steps 5 --warmup 2 --settle-s 0.05 --roofline-only | cut -c1-400
python tools/collect_counters.py --out gpurun_out/${RN}_counters_m8.json --match w4_gemm_xr_kernel --label "m=8 Bint4 4096^2 (w4_gemm_xr_kernel), 512 layers per launch" -- python tools/ab.py 8,4096,4096,1,any4_rowwise,128 | cut -c1-400
python tools/collect_counters.py --out gpurun_out/${RN}_counters_config3.json --match w4_gemm_xr_kernel --label "config 3: m=8, 8192^2, weights on the left in the native format (TG_WFMT_ROWS; w4_gemm_xr_kernel, packed rows), 128 layers per launch" -- python bench.py --steps 5 --warmup 2 --settle-s 0.05 --roofline-only --left --m 8 --n 8192 --k 8192 --layers 128 | cut -c1-400
python tools/collect_counters.py --out gpurun_out/${RN}_counters_config3_reference_words.json --match w4_gemm_pair_kernel --label "config 3 on the reference's Aint4 words: m=8 8192^2, 128 layers per launch" -- python tools/ab.py 8,8192,8192,0,any4_rowwise,128 | cut -c1-400
python tools/collect_counters.py --out gpurun_out/${RN}_counters_m16.json --match w4_gemm_xr_kernel --label "m=16 Bint4 4096^2 (w4_gemm_xr_kernel), 512 layers per launch" -- python tools/ab.py 16,4096,4096,1,any4_rowwise,128 | cut -c1-400
python tools/collect_counters.py --out gpurun_out/${RN}_counters_reference_m1.json --match w4_gemm_stream --label "TG_NUM_REFERENCE m=1 Bint4 4096^2, 512 layers per launch" -- env ANY4_AB_NUMERICS=reference python tools/ab.py 1,4096,4096,1,any4_rowwise,128 | cut -c1-400
echo "== quick_bench default dispatch"; timeout 600 python tools/quick_bench.py --configs "1,4096,4096,1;2,4096,4096,1;4,4096,4096,1;8,4096,4096,1;16,4096,4096,1;1,8192,8192,1;8,8192,8192,1;16,8192,8192,1;1,14336,4096,1;1,4096,14336,1;8,4096,14336,1;16,4096,14336,1;1,4096,4096,0;8,4096,4096,0;1,8192,8192,0;8,8192,8192,0;16,8192,8192,0" --L 256 --iters 3 2>&1 | grep -E "^m=|eager|graph|stacked|steady" | tee gpurun_out/${RN}_quick_bench_default.txt
for q in "int4" "any4_global" "mx4" "int4 --g 32" "any4_rowwise --g 64" "any4_rowwise --g 256"; do timeout 300 python tools/quick_bench.py --configs "1,4096,4096,1;8,4096,4096,1;8,8192,8192,0" --qtype $q --L 256 --iters 3 2>&1 | grep -E "^m=|stacked|steady"; done | tee gpurun_out/${RN}_quick_bench_variants.txt

# the decode step (BASELINE config 5, TP = 1): kernel stats and the per-node timeline of the graph replay
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dec_prof -o dec -- python $R/tools/llama_decode_bench.py --steps 30 --warmup 5 --max-seq 1024 --start-pos 136 --interleave > $R/gpurun_out/${RN}_decode_bench.log 2>&1
cd $R
cp $(find /tmp/dec_prof -name 'dec_kernel_stats.csv' | head -1) gpurun_out/${RN}_decode_kernel_stats.csv
python tools/decode_timeline.py "$(find /tmp/dec_prof -name 'dec_kernel_trace.csv' | head -1)" --last 10 > gpurun_out/${RN}_decode_timeline.txt 2>&1
tail -2 gpurun_out/${RN}_decode_bench.log | cut -c1-300; cat gpurun_out/${RN}_decode_timeline.txt
[ -x variants/graph_chain ] && variants/graph_chain > gpurun_out/${RN}_ubench_graph_chain.txt 2>&1; [ -x variants/overlap_chain ] && timeout 60 variants/overlap_chain > gpurun_out/${RN}_ubench_overlap_chain.txt 2>&1
# (hipcc --offload-arch=gfx950 -O3 -o variants/<name> tools/ubench/<name>.hip beforehand: the binaries are git-ignored but travel)
[ -x variants/grid_barrier ] && timeout 120 variants/grid_barrier > gpurun_out/${RN}_ubench_grid_barrier.txt 2>&1
[ -x variants/fused_qkv_attn ] && timeout 120 variants/fused_qkv_attn > gpurun_out/${RN}_ubench_fused_qkv_attn.txt 2>&1
[ -x variants/lds_store ] && timeout 120 variants/lds_store > gpurun_out/${RN}_ubench_lds_store.txt 2>&1
[ -x variants/persistent_chain ] && timeout 120 variants/persistent_chain 16 > gpurun_out/${RN}_ubench_persistent_chain.txt 2>&1
[ -x variants/x_broadcast ] && timeout 120 variants/x_broadcast > gpurun_out/${RN}_ubench_x_broadcast.txt 2>&1
# the reference's module-level protocol (microbenchmark.py)
timeout 600 python tools/microbenchmark.py --input-dim 4096 --output-dim 4096 --quantize anyq > /dev/null 2>&1   # (the first process on a fresh box reads ~5 us high in its wall-clock column: discarded)
(for k in 4096 8192; do for q in "anyq" "intq" "anyq --quantize-args per_row=False" "anyq"; do echo "##### K=$k --quantize $q"; timeout 600 python tools/microbenchmark.py --input-dim $k --output-dim $k --quantize $q 2>&1 | grep -v "amdgpu.ids\|ROCTracer" | tail -5; done; done) > gpurun_out/${RN}_microbenchmark.txt 2>&1
# seeded random shapes through the ops against the oracle; batched decode steps
timeout 300 python tests/stress_random.py --cases 80 --seed 3 2>&1 | grep -E "bad=[1-9]|TOTAL" > gpurun_out/${RN}_stress_random_summary.txt
(for bs in 4 8 16; do echo -n "bs $bs: "; timeout 300 python tools/llama_decode_bench.py --config llama3_8b --bs $bs --steps 20 --warmup 5 --max-seq 1024 --start-pos 136 --interleave 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['any4']['ms_per_token'] if 'any4' in d else d)"; done) > gpurun_out/${RN}_decode_batched.txt 2>&1
# many activation rows through the modules (tile GEMM, split-K launches) against nn.Linear; the HF protocol at prefill lengths
timeout 600 python dev/many_rows_bench.py 2>&1 | grep "^m=" > gpurun_out/${RN}_many_rows_modules.txt
(for sl in 512 128; do echo "##### seqlen $sl (own kernels, eager)"; timeout 600 python tools/hf_benchmark.py --arch llama3_8b --layers 2 --seqlen $sl 2>&1 | grep -E "Model:|Speedup"; done) > gpurun_out/${RN}_hf_benchmark_prefill.txt 2>&1
# PMC counters of the tile GEMM (m = 512 and the split-K launch at m = 128)
python tools/collect_counters.py --out gpurun_out/${RN}_counters_tile_m128_splitk.json --match w4_gemm_tile_kernel --label "tile GEMM, split-K launch: m=128 4096^2 (64 tiles x 4 splits)" -- python dev/many_rows_bench.py --shapes "128,4096,4096" --layers 6 | cut -c1-300
# the side formats: Int8Linear and the 16-bit-weight tinygemm op against nn.Linear, one 4096^2 layer per graph node
timeout 600 python tools/side_formats_bench.py 2>&1 | grep -E "^int8|^f16" > gpurun_out/${RN}_int8_f16_modules.txt
