set -u
OUT=gpurun_out/r06_e
mkdir -p $OUT
timeout 900 python tools/microbench/gemm_continuous_ab.py > $OUT/gemm_defer_ab.log 2>&1; cat $OUT/gemm_defer_ab.log
timeout 900 python tools/microbench/gemm_continuous_ab.py --fp8 > $OUT/gemm_defer_ab_fp8.log 2>&1; cat $OUT/gemm_defer_ab_fp8.log
python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm" > $OUT/pytest_gemm.log 2>&1; tail -3 $OUT/pytest_gemm.log
python -m pytest tests/test_gpu_dit.py tests/test_gpu_fp8.py -x -q > $OUT/pytest_dit.log 2>&1; tail -3 $OUT/pytest_dit.log
for i in 1 2; do
for t in 1 0; do
PE_GEMM_DEFER_EPILOGUE=$t python bench.py --steps 2 --warmup 1 --no-secondary --no-prologue --no-probes --no-cpu-baseline --no-self-check > $OUT/bench_defer${t}_$i.json 2> $OUT/bench_defer${t}_$i.err
python - $OUT/bench_defer${t}_$i.json <<'P'
import json,sys
j=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
print(sys.argv[1], j["ms_per_step"], j["roofline"]["frac"])
P
done
done
for t in 1 0; do
PE_GEMM_DEFER_EPILOGUE=$t python bench.py --fp8 --steps 2 --warmup 1 --no-secondary --no-prologue --no-probes --no-cpu-baseline --no-self-check > $OUT/bench_fp8_defer${t}.json 2> $OUT/bench_fp8_defer${t}.err
python - $OUT/bench_fp8_defer${t}.json <<'P'
import json,sys
j=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
print(sys.argv[1], j["ms_per_step"], j["roofline"]["frac"])
P
done
