mkdir -p gpurun_out/r05j; o=gpurun_out/r05j
python -m pytest tests -m gpu -x -q > $o/tests.log 2>&1; grep -n "passed\|failed\|Error" $o/tests.log | tail -5
python scripts/api_frame_times.py > $o/api.log 2>&1; tail -4 $o/api.log
for i in 1 2; do python bench.py --mode batch 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch', d.get('frames_per_s'), d.get('frames_per_s_each_round'))"; done
