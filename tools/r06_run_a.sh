set -u
OUT=gpurun_out/r06_a
mkdir -p $OUT
python -m pytest tests/test_gpu_dit.py -x -q -k "trim or eligen or G5 or fp8_attention or controlnet or hot_lora" > $OUT/pytest_dit.log 2>&1; tail -5 $OUT/pytest_dit.log
python -m pytest tests/test_gpu_facade.py -x -q -k "graph_decoder or c1 or c3" > $OUT/pytest_facade.log 2>&1; tail -3 $OUT/pytest_facade.log
for i in 1 2; do
for t in 1 0; do
PE_DEBUG="dit_trim_last_block=$t" python bench.py --steps 2 --warmup 1 --no-secondary --no-prologue --no-probes --no-cpu-baseline --no-self-check > $OUT/bench_trim${t}_$i.json 2> $OUT/bench_trim${t}_$i.err
python - $OUT/bench_trim${t}_$i.json <<'P'
import json,sys
j=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
print(sys.argv[1], j["ms_per_step"], j["roofline"]["frac"], j["whole_path"])
P
done
done
