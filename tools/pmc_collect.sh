#!/bin/bash
# PMC passes over the HEADLINE kernels and shapes: the 60-layer bench command with 2 denoising steps (4 forwards = 240 launches
# of every block kernel; per-launch counters do not depend on the step count).  Separate passes as MI355X_MICROARCH.md
# prescribes (FETCH_SIZE and WRITE_SIZE do not fit one pass; --pmc is never combined with the sys/hip/hsa trace domains).
# usage (GPU box, from the repo root):  bash tools/pmc_collect.sh gpurun_out/pmc_r04 [extra bench flags]
set -u
OUT=${1:-gpurun_out/pmc_r04}; shift || true
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
CMD="python bench.py --layers 60 --inference-steps 2 --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --no-probes --single-stream $*"
mkdir -p "$OUT"
echo "$CMD" > "$OUT/command.txt"
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -o p -- $CMD > "$OUT/$name.log" 2>&1; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run l2 TCC_HIT_sum TCC_MISS_sum
run sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run grbm GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS
find "$OUT" -name "*_counter_collection.csv" | head
