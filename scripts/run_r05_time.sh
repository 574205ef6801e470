t0=$(date +%s.%N); python bench.py --gpus 1 --steps 20 --warmup 3 > /tmp/b.json 2> /tmp/b.err; t1=$(date +%s.%N); echo "bench wall $(echo "$t1 - $t0" | bc) s"
python -c "
import json; d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['steps'], d['roofline']['frac'], d['cpu_baseline']['value'], d['api_multi_device'].get('ms'), d['content']['noise'].get('Mpixel/s'))"
t0=$(date +%s.%N); python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 3 --no-legs --no-cpu-baseline --no-api 2>&1 | tail -1 | cut -c1-160; t1=$(date +%s.%N); echo "torchrun wall $(echo "$t1 - $t0" | bc) s"
