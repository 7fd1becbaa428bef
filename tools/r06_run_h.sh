set -u
OUT=gpurun_out/r06_h
mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "decode" > $OUT/pytest_decode.log 2>&1; tail -3 $OUT/pytest_decode.log
for cfg in "decode_layer_wgs_per_cu=4" "decode_layer_wgs_per_cu=2" "decode_layer_wgs_per_cu=8" "decode_layer_wgs_per_cu=4,decode_layer_no_barrier=1" "decode_layer_wgs_per_cu=2,decode_layer_no_barrier=1"; do
PE_DEBUG="$cfg" PE_DECODE_LAYER_KERNEL=1 timeout 600 python tools/prologue_time.py --quick --decode-tokens 128 > $OUT/prologue_tmp.json 2> $OUT/prologue_tmp.err; echo "$cfg: $(grep -h 'decode_tokens_per_second\|sha1' $OUT/prologue_tmp.json | tr -d '\n')"
done
PE_DECODE_LAYER_KERNEL=0 timeout 600 python tools/prologue_time.py --quick --decode-tokens 128 > $OUT/prologue_layer0.json 2> $OUT/prologue_layer0.err; echo "eight launches per layer: $(grep -h 'decode_tokens_per_second\|sha1' $OUT/prologue_layer0.json | tr -d '\n')"
