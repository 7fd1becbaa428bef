set -u
OUT=gpurun_out/r06_f
mkdir -p $OUT
hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_rate tools/microbench/valu_rate.hip > /dev/null 2>&1 && /tmp/valu_rate > $OUT/valu_rate.log 2>&1; cat $OUT/valu_rate.log
python -m pytest tests/test_gpu_kernels.py -x -q -k "flash_attn_fold" > $OUT/pytest_attn.log 2>&1; tail -3 $OUT/pytest_attn.log
python tools/microbench/attn_ab.py 5,9 > $OUT/attn_ab_v9.log 2>&1; cat $OUT/attn_ab_v9.log
python tools/microbench/attn_after_gemm.py 5,9 > $OUT/attn_after_gemm_v9.log 2>&1; cat $OUT/attn_after_gemm_v9.log
python -m pytest tests/test_gpu_vae.py -x -q > $OUT/pytest_vae.log 2>&1; tail -3 $OUT/pytest_vae.log
python tools/microbench/vae_attn_time.py > $OUT/vae_attn_time.log 2>&1; cat $OUT/vae_attn_time.log
for i in 1 2; do
for v in 9 5; do
python bench.py --attn-variant $v --steps 2 --warmup 1 --no-secondary --no-prologue --no-probes --no-cpu-baseline --no-self-check > $OUT/bench_attn${v}_$i.json 2> $OUT/bench_attn${v}_$i.err
python - $OUT/bench_attn${v}_$i.json <<'P'
import json,sys
j=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
print(sys.argv[1], j["ms_per_step"], j["roofline"]["frac"], (j.get("other_kernels") or {}).get("flash_attn"))
P
done
done
