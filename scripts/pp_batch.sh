FUZZ_BUDGET_S=420 python scripts/fuzz_api_parity.py 4000 20260929 2>&1 | tail -4
FUZZ_BUDGET_S=300 python scripts/fuzz_api_parity.py 400 777 large 2>&1 | tail -4
HYDAMD_DEVICES=0,0,0 FUZZ_BUDGET_S=200 python scripts/fuzz_api_parity.py 300 4242 large 2>&1 | tail -4
