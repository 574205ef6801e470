python -m pytest tests/test_gpu_device_parity.py tests/test_gpu_full_size.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -3
for f in 4 5 6; do echo "form $f:"; python scripts/one_frame.py 3 $f 2 times 2>&1 | grep -E "rans_encode|pack_sections"; done
pp() { GPU_MAX_HW_QUEUES=22 timeout 300 python scripts/pipe_probe.py --reps 1 --frames 512 --streams 16 --batch 2 "$@" 2>&1 | grep -E "SUSTAINED" | cut -c60-400; }
for r in 1 2; do echo -n "loop form 5: "; pp --rans 5; done
