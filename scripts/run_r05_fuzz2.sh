#!/bin/bash
# a second helping of seeded cases on the round's final tree ($1 = commit), other seeds
mkdir -p gpurun_out/r05fuzz2; o=gpurun_out/r05fuzz2/fuzz2.txt
echo "# second sweep, commit $1" > $o
python scripts/fuzz_api_parity.py 30000 93001 2>&1 | grep -v amdgpu >> $o
FUZZ_BUDGET_S=500 python scripts/fuzz_api_parity.py 700 93002 large 2>&1 | grep -v amdgpu >> $o
echo "# HYDAMD_DEVICES=0,0 HYDAMD_VERIFY_PEERS=1" >> $o
HYDAMD_DEVICES=0,0 HYDAMD_VERIFY_PEERS=1 FUZZ_BUDGET_S=300 python scripts/fuzz_api_parity.py 500 93003 large 2>&1 | grep -v amdgpu >> $o
echo "# HYDAMD_TILE_PIPELINE=4 GPU_MAX_HW_QUEUES=22" >> $o
GPU_MAX_HW_QUEUES=22 HYDAMD_TILE_PIPELINE=4 python scripts/fuzz_api_parity.py 8000 93004 2>&1 | grep -v amdgpu >> $o
cat $o
