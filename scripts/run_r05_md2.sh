python -m pytest tests/test_gpu_device_parity.py tests/test_gpu_api_parity.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -1
bash scripts/run_r05_md.sh | grep -E "RESULT|join|transform_tok" | cut -c1-140
python scripts/api_tile_mode.py 4096 8 2>&1 | grep "shift  0"
python scripts/fuzz_api_parity.py 6000 95001 | tail -1
