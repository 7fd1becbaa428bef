#!/bin/bash
# The round's evidence set on ONE box (GPU box, from the repo root):  bash tools/final_run.sh gpurun_out/<dir>
#   pytest.log          python -m pytest tests -m gpu --durations=25
#   bench_default.json  the default bench line as the driver runs it (two streams, self-check, secondary configs, prologue, probes, cpu_baseline)
#   kernel_stats.md / pmc.json   tools/evidence_run.sh: rocprofv3 --kernel-trace --stats summary + PMC passes of the single-stream form
set -u
OUT=${1:-gpurun_out/final}
mkdir -p "$OUT"
python -m pytest tests -m gpu -q --durations=25 > "$OUT/pytest.log" 2>&1
tail -3 "$OUT/pytest.log"
python bench.py --steps ${BENCH_STEPS:-5} --warmup 1 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
bash tools/evidence_run.sh "$OUT"
# the e4m3 attention operator per kernel variant, and the VALU / MFMA co-issue probe behind profiles/r04_attention_notes.md section 5.2
python tools/microbench/attn_fp8_time.py > "$OUT/attn_fp8_time.log" 2>&1
hipcc --offload-arch=gfx950 -O3 -o /tmp/coissue tools/microbench/mfma_f8_coissue.hip > /dev/null 2>&1 && /tmp/coissue > "$OUT/mfma_f8_coissue.log" 2>&1
python - "$OUT" <<'P'
import json, sys
o = sys.argv[1]
j = json.loads([l for l in open(o + "/bench_default.json").read().splitlines() if l.startswith("{")][-1])
print("ms_per_step", j["ms_per_step"], "gemm", j["roofline"]["achieved"], j["roofline"]["frac"], "determinism", j.get("determinism"))
for k, v in (j.get("secondary") or {}).items():
    print(k, v.get("ms_per_image"))
print(j.get("other_kernels"))
print(j.get("prologue"))
P
