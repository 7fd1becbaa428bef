set -u
OUT=gpurun_out/r06_b
mkdir -p $OUT
python tools/microbench/gemm_epilogue_cost.py > $OUT/gemm_epilogue_cost.log 2>&1; cat $OUT/gemm_epilogue_cost.log
python tools/microbench/gemm_epilogue_cost.py --fp8 > $OUT/gemm_epilogue_cost_fp8.log 2>&1; cat $OUT/gemm_epilogue_cost_fp8.log
python -m pytest tests/test_gpu_dit.py -x -q -k "trim or eligen" > $OUT/pytest_dit.log 2>&1; tail -3 $OUT/pytest_dit.log
