set -u
OUT=gpurun_out/r06_n
mkdir -p $OUT
python -m pytest tests/test_gpu_dit.py tests/test_gpu_fp8.py tests/test_gpu_kernels.py -x -q -k "fp8 or qkv or e4m3 or G5 or statistics" > $OUT/pytest_fp8.log 2>&1; tail -4 $OUT/pytest_fp8.log
python -m pytest tests/test_gpu_facade.py -x -q -k "c3 or c1" > $OUT/pytest_facade.log 2>&1; tail -2 $OUT/pytest_facade.log
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
python tools/microbench/attn_fp8_time.py > $OUT/attn_fp8_time.log 2>&1; tail -4 $OUT/attn_fp8_time.log
