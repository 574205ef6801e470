for r in 1 2 3; do for f in 5 6; do
echo -n "form $f: "; python bench.py --rans-waves $f --no-cpu-baseline --no-api 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['timing']['Mpixel/s_each_window'], 'batch_dev', d['batch_4k_device']['frames_per_s'], 'shard', d['shard_16k']['ms_per_step'], 'B1', d['one_frame_per_launch_group']['Mpixel/s'])"
done; done
