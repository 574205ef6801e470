mkdir -p gpurun_out/r05r; export TMPDIR=/tmp; root=$PWD
cd /tmp && rm -rf /tmp/kt_tile && rocprofv3 --kernel-trace --memory-copy-trace --stats -d /tmp/kt_tile -o kt -- python $root/scripts/tile_trace_client.py > /tmp/kt_tile.log 2>&1
cd $root; db=$(find /tmp/kt_tile -name "*.db" | head -1); python scripts/rocpd_timeline.py $db 44 > gpurun_out/r05r/tile_timeline.txt 2>&1; cat gpurun_out/r05r/tile_timeline.txt
