python -m pytest tests/test_gpu_device_parity.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -3
for f in 4; do echo "form $f, operands a chunk ahead:"; python scripts/one_frame.py 3 $f 2 times 2>&1 | grep -E "rans_encode"; done
