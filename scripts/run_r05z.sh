mkdir -p gpurun_out/r05z; o=gpurun_out/r05z
python -m pytest tests -m gpu -x -q > $o/tests.log 2>&1; grep -n "passed\|failed\|Error" $o/tests.log | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python scripts/fuzz_api_parity.py 8000 90001 2>&1 | tail -1
FUZZ_BUDGET_S=120 python scripts/fuzz_api_parity.py 200 90002 large 2>&1 | tail -1
