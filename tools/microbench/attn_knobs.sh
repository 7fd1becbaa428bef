#!/bin/bash
# Builds one library per schedule-knob set of the folded attention body (tools/gen_attn_w4.py env knobs) into build_ab/ (HERE, no GPU
# needed), for `attn_knobs_run.sh` to time on the GPU box.  Usage: tools/microbench/attn_knobs.sh "tag:ENV=V ENV2=V" ...
set -eu
cd "$(dirname "$0")/../.."
mkdir -p build_ab
FLAGS="--offload-arch=gfx950 -O3 -std=c++20 -fPIC -ffp-contract=off -Wno-unused-function"
OBJS=$(ls physicedit_amd/_build/*.o | grep -v attention.o)
for spec in "$@"; do
    tag="${spec%%:*}"; envs="${spec#*:}"
    env $envs W4_ONLY=w5 python tools/gen_attn_w4.py > /dev/null
    extra=""; case "$envs" in *W4_STAMPS=1*) extra="-DPE_W4_STAMPS=1";; esac
    /opt/rocm/bin/hipcc $FLAGS $extra -c physicedit_amd/csrc/attention.hip -o build_ab/attention_$tag.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_ab/libpe_$tag.so $OBJS build_ab/attention_$tag.o
    rm build_ab/attention_$tag.o
    echo "built build_ab/libpe_$tag.so  ($envs)"
done
W4_ONLY=w5 python tools/gen_attn_w4.py > /dev/null      # back to the committed defaults
git diff --quiet physicedit_amd/csrc/attention_w5_body.inc || echo "WARNING: attention_w5_body.inc differs from the committed one"
