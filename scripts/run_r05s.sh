mkdir -p gpurun_out/r05s; o=gpurun_out/r05s
python -m pytest tests/test_gpu_api_parity.py tests/test_gpu_fuzz.py tests/test_c_client.py -m gpu -x -q > $o/tests.log 2>&1; grep -n "passed\|failed\|Error" $o/tests.log | tail -5
python scripts/api_tile_mode.py 4096 8 2>&1 | tail -5
HYDAMD_STAGED_READBACK=0 python scripts/api_tile_mode.py 4096 8 2>&1 | tail -5
HYDAMD_TILE_PIPELINE=8 python scripts/api_tile_mode.py 4096 8 2>&1 | tail -5
HYDAMD_TILE_PIPELINE=8 python scripts/fuzz_api_parity.py 2000 61001 | tail -2
python scripts/fuzz_api_parity.py 4000 61002 | tail -2
