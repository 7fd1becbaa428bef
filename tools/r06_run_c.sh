set -u
OUT=gpurun_out/r06_c
mkdir -p $OUT
timeout 600 python tools/microbench/gemm_continuous_ab.py > $OUT/gemm_continuous_ab.log 2>&1; cat $OUT/gemm_continuous_ab.log
timeout 600 python tools/microbench/gemm_continuous_ab.py --fp8 > $OUT/gemm_continuous_ab_fp8.log 2>&1; cat $OUT/gemm_continuous_ab_fp8.log
for i in 1 2; do
for t in 1 0; do
PE_GEMM_CONTINUOUS=$t python bench.py --steps 2 --warmup 1 --no-secondary --no-prologue --no-probes --no-cpu-baseline --no-self-check > $OUT/bench_cont${t}_$i.json 2> $OUT/bench_cont${t}_$i.err
python - $OUT/bench_cont${t}_$i.json <<'P'
import json,sys
j=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
print(sys.argv[1], j["ms_per_step"], j["roofline"]["frac"])
P
done
done
