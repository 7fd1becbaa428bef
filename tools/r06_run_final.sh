#!/bin/bash
# The round's evidence set on ONE box (GPU box, repo root):  bash tools/r06_run_final.sh [out dir]
#   pytest.log, bench_default.json, kernel_stats.md / pmc.json (tools/evidence_run.sh), kernel_stats_fp8.md
set -u
OUT=${1:-gpurun_out/r06_final3}
mkdir -p $OUT
python -m pytest tests -m gpu -q --durations=25 > "$OUT/pytest.log" 2>&1
tail -3 "$OUT/pytest.log"
python bench.py --steps 5 --warmup 1 > $OUT/bench_default.json 2> $OUT/bench_default.err
python - "$OUT" <<'P'
import json, sys
o = sys.argv[1]
j = json.loads([l for l in open(o + "/bench_default.json").read().splitlines() if l.startswith("{")][-1])
print("ms_per_step", j["ms_per_step"], "gemm", j["roofline"]["achieved"], j["roofline"]["frac"], "determinism", j.get("determinism"))
for k, v in (j.get("secondary") or {}).items():
    print(k, v.get("ms_per_image"))
print({k: v for k, v in (j.get("prologue") or {}).items() if "token" in k or "TBps" in k or "frac" in k})
print(j.get("whole_path"))
P
if [ "${SKIP_PROFILES:-0}" = "0" ]; then
bash tools/evidence_run.sh "$OUT"
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -d "$OUT/prof_fp8" -o bench -- python bench.py --fp8 --fp8-attention --single-stream --no-cpu-baseline --no-secondary --no-probes --no-prologue --no-self-check --steps 1 --warmup 0 > "$OUT/prof_fp8_bench.json" 2> "$OUT/prof_fp8_bench.err"
DB=$(find "$OUT/prof_fp8" -name "*_results.db" | head -1)
python tools/rocprof_summary.py "$DB" "$OUT/kernel_stats_fp8.md" > /dev/null
rm -rf "$OUT/prof_fp8"
head -14 "$OUT/kernel_stats_fp8.md"
fi
