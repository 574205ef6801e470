python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed" | tail -1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
HYDAMD_DEVICES=0,0,0 HYDAMD_VERIFY_PEERS=1 FUZZ_BUDGET_S=200 python scripts/fuzz_api_parity.py 300 97003 large 2>&1 | tail -1
FUZZ_BUDGET_S=300 python scripts/fuzz_api_parity.py 600 97004 large 2>&1 | tail -1
python scripts/fuzz_api_parity.py 10000 97005 2>&1 | tail -1
HYDAMD_TILE_PIPELINE=8 python scripts/fuzz_api_parity.py 4000 97006 2>&1 | tail -1
