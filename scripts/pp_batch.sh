pp() { GPU_MAX_HW_QUEUES=22 timeout 300 python scripts/pipe_probe.py --reps 1 --frames 512 --streams 32 "$@" 2>&1 | grep -E "SUSTAINED|stage times" | cut -c60-400; }
python -m pytest tests/test_gpu_device_parity.py tests/test_gpu_full_size.py -x -q -m gpu 2>&1 | tail -3
for r in 1 2; do for f in 5 6; do echo -n "form $f B=2: "; pp --batch 2 --rans $f; done; done
python bench.py 2>gpurun_out/bench.err | tail -1 > gpurun_out/bench.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench.json"))
print(d["value"], d["timing"], d.get("value_by_the_method_of_rounds_1_to_3"))
for k in ("one_frame_per_launch_group","hf_sections_only","finished_file_per_step","batch_4k_device","batch_4k","shard_16k","single_frame_form5"):
    print(k, json.dumps(d["config"].get(k) if k in d.get("config",{}) else d.get(k))[:400])
PY
