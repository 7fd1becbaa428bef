#!/bin/bash
# On the GPU box: time every build_ab/libpe_*.so (variant 5 and the variant-4 control of the same library), two rounds.
cd "$(dirname "$0")/../.."
for round in 1 2; do
    for lib in build_ab/libpe_*.so; do
        echo "== $lib (round $round)"
        PE_LIB_PATH=$PWD/$lib python tools/microbench/attn_ab.py 5,4 2>&1 | grep "^S="
    done
done
