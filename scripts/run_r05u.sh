mkdir -p gpurun_out/r05u; o=gpurun_out/r05u/r05_tile_mode.txt
{
echo "# tile mode through the drop-in API, 4096x4096 RGB8 photo (scripts/api_tile_mode.py 4096 8); shift -1 = one-frame mode for comparison"
echo "## default: every hyd_send_tile call ends with its tile's frame, as in the reference (one frame at a time)"
python scripts/api_tile_mode.py 4096 8 2>&1 | grep shift
echo "## the same with the separate read-backs of round 4 (HYDAMD_STAGED_READBACK=0)"
HYDAMD_STAGED_READBACK=0 python scripts/api_tile_mode.py 4096 8 2>&1 | grep shift
echo "## eight tile frames in flight (HYDAMD_TILE_PIPELINE=8 / hydamd_set_tile_pipeline), GPU_MAX_HW_QUEUES=22"
GPU_MAX_HW_QUEUES=22 HYDAMD_TILE_PIPELINE=8 python scripts/api_tile_mode.py 4096 8 2>&1 | grep shift
echo "## ... with the separate read-backs (HYDAMD_STAGED_READBACK=0)"
GPU_MAX_HW_QUEUES=22 HYDAMD_TILE_PIPELINE=8 HYDAMD_STAGED_READBACK=0 python scripts/api_tile_mode.py 4096 8 2>&1 | grep shift
} > $o 2>&1
cat $o
