HYDAMD_TRACE=1 HYDAMD_DEVICES=0,0,0 HYDAMD_VERIFY_PEERS=1 FUZZ_BUDGET_S=200 python scripts/fuzz_api_parity.py 200 97003 large 2>&1 | grep -v "ms$" | grep -i "device error\|mismatch\|Error\|cases" | head -8
echo "--- without verify"
HYDAMD_TRACE=1 HYDAMD_DEVICES=0,0,0 FUZZ_BUDGET_S=200 python scripts/fuzz_api_parity.py 200 97003 large 2>&1 | grep -i "device error\|mismatch\|Error\|cases" | head -8
