#!/bin/bash
# rocprofv3 kernel-trace summary (single-stream form of the headline workload, one image) + the PMC passes, on one box:
#   bash tools/evidence_run.sh gpurun_out/<dir>      ->  <dir>/kernel_stats.md, <dir>/pmc.json   (raw outputs deleted: 64 MiB limit)
set -u
OUT=${1:-gpurun_out/evidence}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o bench -- python bench.py --single-stream --no-cpu-baseline --no-secondary --no-probes \
    --no-prologue --no-self-check --steps 1 --warmup 0 > "$OUT/prof_bench.json" 2> "$OUT/prof_bench.err"
DB=$(find "$OUT/prof" -name "*_results.db" | head -1)
python tools/rocprof_summary.py "$DB" "$OUT/kernel_stats.md" > /dev/null
rm -rf "$OUT/prof"
bash tools/pmc_collect.sh "$OUT/pmc" --no-prologue --no-self-check > "$OUT/pmc.log" 2>&1
python tools/pmc_summary.py "$OUT/pmc" "$OUT/pmc.json" >> "$OUT/pmc.log" 2>&1
rm -rf "$OUT/pmc"
head -14 "$OUT/kernel_stats.md"
