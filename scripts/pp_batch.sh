python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc $?"; grep -E "passed|failed|error" gpurun_out/pytest_gpu.log | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py 2>gpurun_out/bench.err | tail -1 > gpurun_out/bench.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench.json"))
print(d["value"], d["timing"]["Mpixel/s_each_window"], d.get("value_by_the_method_of_rounds_1_to_3"), d["roofline"]["frac"], d["roofline_transform_kernel"])
for k in ("one_frame_per_launch_group","hf_sections_only","finished_file_per_step","batch_4k_device","batch_4k","shard_16k","single_frame","single_frame_form5","api_end_to_end"):
    print(k, json.dumps(d["config"].get(k) if k in d.get("config",{}) else d.get(k))[:260])
PY
