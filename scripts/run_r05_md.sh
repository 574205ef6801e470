mkdir -p gpurun_out/r05md; export TMPDIR=/tmp; root=$PWD
cd /tmp && rm -rf /tmp/kt_md && HYDAMD_DEVICES=0,0,0,0 rocprofv3 --kernel-trace --memory-copy-trace --stats -d /tmp/kt_md -o kt -- python $root/scripts/api_multi_device_client.py > /tmp/kt_md.log 2>&1
cd $root; db=$(find /tmp/kt_md -name "*.db" | head -1)
{ echo "# HYDAMD_DEVICES=0,0,0,0 python scripts/api_multi_device_client.py under rocprofv3 --kernel-trace: six 16384x16384 RGB8 frames through hyd_send_tile, the frame dealt to four contexts of one GPU"; grep RESULT /tmp/kt_md.log; python scripts/rocpd_summary.py $db | grep -v "at::native"; echo; echo "## the last frame's closing stage (kernels and copies in start order)"; python scripts/rocpd_timeline.py $db 60; } > gpurun_out/r05md/r05_kernel_stats_multi_device.txt 2>&1
head -30 gpurun_out/r05md/r05_kernel_stats_multi_device.txt | cut -c1-150
