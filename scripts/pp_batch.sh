FUZZ_BUDGET_S=500 python scripts/fuzz_api_parity.py 20000 99001 2>&1 | tail -2
FUZZ_BUDGET_S=400 python scripts/fuzz_api_parity.py 600 31337 large 2>&1 | tail -2
HYDAMD_TILE_PIPELINE=8 FUZZ_BUDGET_S=250 python scripts/fuzz_api_parity.py 5000 555 2>&1 | tail -2
HYDAMD_RANS_WAVES=5 FUZZ_BUDGET_S=200 python scripts/fuzz_api_parity.py 5000 808 2>&1 | tail -2
HYDAMD_RANS_WAVES=6 FUZZ_BUDGET_S=200 python scripts/fuzz_api_parity.py 5000 909 2>&1 | tail -2
