mkdir -p gpurun_out/r05i; o=gpurun_out/r05i
python -m pytest tests -m gpu -x -q > $o/tests.log 2>&1; grep -n "passed\|failed\|Error" $o/tests.log | tail -5
python bench.py > $o/bench.json 2> $o/bench.err; tail -c 600 $o/bench.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r05i/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['timing']['Mpixel/s_each_window'], d['timed_contexts_as_files'])
for k in ('single_frame','single_frame_form5','api_end_to_end','api_multi_device','content','shard_16k','batch_4k_device','batch_4k'):
    print(k, json.dumps(d.get(k))[:900])
P
