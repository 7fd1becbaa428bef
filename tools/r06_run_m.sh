set -u
OUT=gpurun_out/r06_m
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -4 $OUT/smoke.log
python -m pytest tests/test_gpu_dit.py -x -q -k "eligen or G5" > $OUT/pytest_eligen.log 2>&1; tail -2 $OUT/pytest_eligen.log
python bench.py --steps 2 --warmup 1 --no-secondary --no-prologue --no-probes --no-cpu-baseline > $OUT/bench_short.json 2> $OUT/bench_short.err
python - $OUT/bench_short.json <<'P'
import json,sys
j=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
print(round(j["ms_per_step"],1), "ms/image  gemm frac", round(j["roofline"]["frac"],4), "traffic", j["roofline"]["traffic"], j["roofline"]["traffic_source"][:30], "determinism", j["determinism"])
P
