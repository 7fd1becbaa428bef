#!/bin/bash
# The round's evidence set on ONE box (GPU box, from the repo root):  bash tools/final_run.sh gpurun_out/<dir>
#   pytest.log          python -m pytest tests -m gpu --durations=25
#   bench_default.json  the default bench line as the driver runs it (two streams, secondary configs, probes, cpu_baseline)
#   kernel_stats.md     rocprofv3 --kernel-trace --stats summary of the single-stream form of the same workload
#   pmc.json            PMC passes (tools/pmc_collect.sh) summarised
# Raw rocprofv3 outputs are deleted after summarising: gpurun copies back at most 64 MiB.
set -u
OUT=${1:-gpurun_out/final}
mkdir -p "$OUT"
python -m pytest tests -m gpu -q --durations=25 > "$OUT/pytest.log" 2>&1
tail -3 "$OUT/pytest.log"
python bench.py --steps 3 --warmup 1 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o bench -- python bench.py --single-stream --no-cpu-baseline --no-secondary --no-probes --steps 1 --warmup 0 \
    > "$OUT/prof_bench.json" 2> "$OUT/prof_bench.err"
DB=$(find "$OUT/prof" -name "*_results.db" | head -1)
python tools/rocprof_summary.py "$DB" "$OUT/kernel_stats.md"
rm -rf "$OUT/prof"
bash tools/pmc_collect.sh "$OUT/pmc" > "$OUT/pmc.log" 2>&1
python tools/pmc_summary.py "$OUT/pmc" "$OUT/pmc.json" >> "$OUT/pmc.log" 2>&1
rm -rf "$OUT/pmc"
python - "$OUT" <<'P'
import json, sys
o = sys.argv[1]
j = json.loads([l for l in open(o + "/bench_default.json").read().splitlines() if l.startswith("{")][-1])
print("ms_per_step", j["ms_per_step"], "gemm", j["roofline"]["achieved"], j["roofline"]["frac"])
for k, v in (j.get("secondary") or {}).items():
    print(k, v.get("ms_per_image"))
print(j.get("other_kernels"))
P
head -12 "$OUT/kernel_stats.md"
