python -m pytest tests/test_gpu_device_parity.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -3
echo "form 4, deferred emission:"; python scripts/one_frame.py 3 4 1 times 2>&1 | grep -E "rans_encode|pack_sections"
echo "form 4, chain writes its own bits:"; HYDAMD_WAVE_FORM_EMITS=1 python scripts/one_frame.py 3 4 1 times 2>&1 | grep -E "rans_encode|pack_sections"
