mkdir -p gpurun_out/r05v; o=gpurun_out/r05v
python -m pytest tests -m gpu -x -q > $o/tests.log 2>&1; grep -n "passed\|failed\|Error" $o/tests.log | tail -5
python scripts/one_frame.py 3 5 2 t 2>&1 | grep -E "transform|rans"
for s in 1 0 1 0; do echo "== split $s"; HYDAMD_K1_SPLIT=$s python scripts/api_tile_mode.py 4096 8 2>&1 | grep "shift  0\|shift  3"; HYDAMD_K1_SPLIT=$s python scripts/api_frame_times.py 2>&1 | tail -2; done
python scripts/pipe_probe.py --streams 16 --batch 2 --frames 512 --rans 5 --reps 2 2>&1 | grep SUSTAINED
python scripts/fuzz_api_parity.py 8000 71001 | tail -1
FUZZ_BUDGET_S=200 python scripts/fuzz_api_parity.py 300 71002 large | tail -1
HYDAMD_TILE_PIPELINE=8 python scripts/fuzz_api_parity.py 3000 71003 | tail -1
