#!/bin/bash
# Everything under profiles/ for one round, in one go on the GPU box (copy gpurun_out/<tag>/* to profiles/ afterwards):
#   kernel traces (rocprofv3 --kernel-trace --stats) of the default bench line, of single frames, of the 16K shard mode
#   and of the drop-in API; the PMC passes (separate runs, never combined with tracing), the K1 phase probe, the VALU
#   rate table.
# usage: bash scripts/collect_profiles.sh <tag>          e.g. r03
set -u
tag=$1
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
root=$PWD
python bench.py > "$out/${tag}_bench_default.json" 2> "$out/bench_default.err"
cd /tmp
rm -rf /tmp/kt_single /tmp/kt_pipe /tmp/kt_shard /tmp/kt_api
rocprofv3 --kernel-trace --stats -d /tmp/kt_single -o kt -- python "$root/scripts/one_frame.py" 5 5 2 > /tmp/kt_single.log 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/kt_pipe -o kt -- python "$root/bench.py" --steps 256 --no-cpu-baseline --no-api --no-legs > /tmp/kt_pipe.log 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/kt_shard -o kt -- python "$root/bench.py" --mode shard --steps 30 > /tmp/kt_shard.log 2>&1
rocprofv3 --kernel-trace --memory-copy-trace --stats -d /tmp/kt_api -o kt -- python "$root/scripts/api_frame_times.py" > /tmp/kt_api.log 2>&1
cd "$root"
for k in single pipe shard api; do
  db=$(find /tmp/kt_$k -name "*.db" | head -1)
  if [ -n "$db" ]; then python scripts/rocpd_summary.py "$db" | grep -v "at::native" > "$out/${tag}_kernel_stats_$k.txt" 2>&1; else tail -5 /tmp/kt_$k.log > "$out/${tag}_kernel_stats_$k.txt"; fi
done
db=$(find /tmp/kt_api -name "*.db" | head -1)
[ -n "$db" ] && python scripts/rocpd_timeline.py "$db" 75 > "$out/${tag}_api_timeline.txt" 2>&1  # the last frame: uploads, kernels, read-back
python scripts/api_tile_mode.py 4096 8 > "$out/${tag}_tile_mode_now.txt" 2>&1
grep "^{" /tmp/kt_pipe.log | tail -1 > "$out/${tag}_bench_under_rocprof.json"
grep "^{" /tmp/kt_shard.log | tail -1 > "$out/${tag}_shard_under_rocprof.json"
bash scripts/collect_pmc.sh "$out/pmc" python scripts/one_frame.py 3 5 2 > /dev/null 2>&1
cat "$out"/pmc/pmc_set*.txt > "$out/${tag}_pmc_8k_photo.txt"
ls -la "$out"
