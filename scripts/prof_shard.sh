set -u
export TMPDIR=/tmp
root=$PWD
mkdir -p gpurun_out/r3_prof
cd /tmp
rm -rf /tmp/kt_shard
rocprofv3 --kernel-trace --stats -d /tmp/kt_shard -o kt -- python "$root/bench.py" --mode shard --steps 30 --shard-depth 4 > /tmp/kt_shard.log 2>&1
cd "$root"
db=$(find /tmp/kt_shard -name "*.db" | head -1)
python scripts/rocpd_summary.py "$db" > gpurun_out/r3_prof/shard_kernel_stats.txt 2>&1
python scripts/rocpd_gaps.py "$db" 20 > gpurun_out/r3_prof/shard_gaps.txt 2>&1
grep "^{" /tmp/kt_shard.log | tail -1 | cut -c1-300
cat gpurun_out/r3_prof/shard_kernel_stats.txt | head -40
cat gpurun_out/r3_prof/shard_gaps.txt | head -40
