s() { timeout 300 python bench.py --mode shard --steps 60 "$@" 2>gpurun_out/err.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('host_ms_per_frame'), d.get('assembled_frames_identical_to_host_assembly'))"; }
for d in 4 6 8 10 12; do echo -n "depth $d: "; s --shard-depth $d; done
echo -n "depth 8 form 5: "; s --shard-depth 8 --rans-waves 5
echo -n "depth 8 pinned: "; s --shard-depth 8 --assemble device-pinned
