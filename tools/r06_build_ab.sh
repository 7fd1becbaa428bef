#!/bin/bash
# A/B builds of the library that differ in gemm.hip only (CPU, repo root):  bash tools/r06_build_ab.sh
#   build_ab/lib_defer.so         the deferred gated-residual epilogue compiled IN (-DPE_GEMM_DEFER; knob gemm_defer_epilogue then switches it)
#   build_ab/lib_slp.so           the default kernel with SLP vectorisation on for gemm.hip (round 5's flags)
# (the default build: DEFER compiled out, gemm.hip without SLP vectorisation)
set -e
mkdir -p build_ab
F="--offload-arch=gfx950 -O3 -std=c++20 -fPIC -ffp-contract=off -Wno-unused-function"
OBJS=$(ls physicedit_amd/_build/*.o | grep -v "/gemm.o")
/opt/rocm/bin/hipcc $F -fno-slp-vectorize -DPE_GEMM_DEFER -c physicedit_amd/csrc/gemm.hip -o build_ab/gemm_defer.o &
/opt/rocm/bin/hipcc $F -c physicedit_amd/csrc/gemm.hip -o build_ab/gemm_slp.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_ab/lib_defer.so $OBJS build_ab/gemm_defer.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_ab/lib_slp.so $OBJS build_ab/gemm_slp.o
ls -la build_ab/*.so
