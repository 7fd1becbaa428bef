set -u
OUT=gpurun_out/r06_l
mkdir -p $OUT
for r in 1 2 3; do
for l in "" build_ab/lib_r05gemm.so; do
PE_LIB_PATH=$l python tools/microbench/gemm_lib_time.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/gemm_lib_ab_r05.log
done
done
for r in 1 2 3; do
for l in "" build_ab/lib_r05gemm.so; do
PE_LIB_PATH=$l python bench.py --steps 2 --warmup 1 --no-secondary --no-prologue --no-probes --no-cpu-baseline --no-self-check > $OUT/bench_tmp.json 2> $OUT/bench_tmp.err
python - $OUT/bench_tmp.json "${l:-default}" <<'P' | tee -a $OUT/bench_lib_ab_r05.log
import json,sys
j=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
print(sys.argv[2], round(j["ms_per_step"],1), "ms/image  gemm frac", round(j["roofline"]["frac"],4))
P
done
done
