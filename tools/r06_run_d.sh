set -u
OUT=gpurun_out/r06_d
mkdir -p $OUT
timeout 900 python tools/microbench/gemm_continuous_ab.py > $OUT/gemm_defer_ab.log 2>&1; cat $OUT/gemm_defer_ab.log
timeout 900 python tools/microbench/gemm_continuous_ab.py --fp8 > $OUT/gemm_defer_ab_fp8.log 2>&1; cat $OUT/gemm_defer_ab_fp8.log
