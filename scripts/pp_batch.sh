python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 8 --warmup 2 --no-cpu-baseline 2>gpurun_out/torchrun.err | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('torchrun N=1:', d['value'], d['n_gpus'], d['steps'], d['warmup'], d['ms_per_step'], d['scaling'], d['shard_16k']['ms_per_step'], d['batch_4k']['frames_per_s'])"
tail -3 gpurun_out/torchrun.err
python bench.py --steps 5 --warmup 2 --no-legs --no-api --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('steps 5:', d['value'], d['steps'], d['timing']['timed_frames'], d['timing']['Mpixel/s_each_window'])"
