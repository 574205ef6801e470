mkdir -p gpurun_out/r05m; o=gpurun_out/r05m
python -m pytest tests/test_gpu_device_parity.py tests/test_gpu_fuzz.py tests/test_gpu_api_parity.py tests/test_gpu_full_size.py -m gpu -x -q > $o/tests.log 2>&1; grep -n "passed\|failed\|Error" $o/tests.log | tail -5
for i in 1 2; do python scripts/one_frame.py 3 4 2 t 2>&1 | grep -E "rans|rror"; done
python scripts/api_tile_mode.py 4096 8 2>&1 | tail -5
python scripts/api_frame_times.py 2>&1 | tail -3
