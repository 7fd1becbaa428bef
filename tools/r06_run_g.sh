set -u
OUT=gpurun_out/r06_g
mkdir -p $OUT
python -m pytest tests/test_gpu_vae.py -x -q > $OUT/pytest_vae.log 2>&1; tail -3 $OUT/pytest_vae.log
python tools/microbench/vae_attn_time.py > $OUT/vae_attn_time.log 2>&1; cat $OUT/vae_attn_time.log
python -m pytest tests/test_gpu_kernels.py -x -q -k "flash_attn" > $OUT/pytest_attn.log 2>&1; tail -3 $OUT/pytest_attn.log
