set -u
OUT=gpurun_out/r06_o
mkdir -p $OUT
python -m pytest tests/test_gpu_dit.py tests/test_gpu_kernels.py -x -q -k "fp8 or statistics" > $OUT/pytest_fp8.log 2>&1; tail -2 $OUT/pytest_fp8.log
for r in 1 2; do
for k in 1 0; do
PE_DEBUG="dit_qkv_stats=$k" python bench.py --fp8 --fp8-attention --steps 2 --warmup 1 --no-secondary --no-prologue --no-probes --no-cpu-baseline --no-self-check > $OUT/bench_tmp.json 2> $OUT/bench_tmp.err
python - $OUT/bench_tmp.json "dit_qkv_stats=$k" <<'P' | tee -a $OUT/bench_fp8attn_stats_ab.log
import json,sys
j=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
print(sys.argv[2], round(j["ms_per_step"],1), "ms/image (e4m3 Linears + e4m3 attention)")
P
done
done
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -d "$OUT/prof_fp8" -o bench -- python bench.py --fp8 --fp8-attention --single-stream --no-cpu-baseline --no-secondary --no-probes --no-prologue --no-self-check --steps 1 --warmup 0 > "$OUT/prof_fp8_bench.json" 2> "$OUT/prof_fp8_bench.err"
DB=$(find "$OUT/prof_fp8" -name "*_results.db" | head -1)
python tools/rocprof_summary.py "$DB" "$OUT/kernel_stats_fp8.md" > /dev/null
rm -rf "$OUT/prof_fp8"
head -16 "$OUT/kernel_stats_fp8.md" | cut -c1-200
