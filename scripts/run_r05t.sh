export GPU_MAX_HW_QUEUES=22
for s in 1 0 1 0; do echo "== staged $s, pipeline 8"; HYDAMD_STAGED_READBACK=$s HYDAMD_TILE_PIPELINE=8 python scripts/api_tile_mode.py 4096 8 2>&1 | tail -4; done
