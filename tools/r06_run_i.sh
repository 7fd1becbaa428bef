set -u
OUT=gpurun_out/r06_i
mkdir -p $OUT
python -m pytest tests/test_gpu_parity_configs.py -x -q -s -k "40_step or two_cfg" > $OUT/pytest_g25.log 2>&1; grep "parity\]\|passed\|failed\|Error" $OUT/pytest_g25.log | cut -c1-400
