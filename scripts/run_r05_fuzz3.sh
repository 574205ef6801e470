#!/bin/bash
# the overflow-rerun paths under fuzz: token and payload buffers so small that most frames outgrow them
export HYDAMD_TOKEN_CAP=4096 HYDAMD_PAYLOAD_CAP=16384
mkdir -p gpurun_out/r05fuzz3; o=gpurun_out/r05fuzz3/fuzz3.txt
echo "# HYDAMD_TOKEN_CAP=4096 HYDAMD_PAYLOAD_CAP=16384 (most frames are rerun with larger buffers), commit $1" > $o
python scripts/fuzz_api_parity.py 6000 98001 2>&1 | grep -v amdgpu | tail -3 >> $o
FUZZ_BUDGET_S=240 python scripts/fuzz_api_parity.py 400 98002 large 2>&1 | grep -v amdgpu | tail -3 >> $o
echo "# ... HYDAMD_TILE_PIPELINE=8" >> $o
HYDAMD_TILE_PIPELINE=8 python scripts/fuzz_api_parity.py 3000 98003 2>&1 | grep -v amdgpu | tail -3 >> $o
echo "# ... HYDAMD_DEVICES=0,0,0 HYDAMD_VERIFY_PEERS=1" >> $o
HYDAMD_DEVICES=0,0,0 HYDAMD_VERIFY_PEERS=1 FUZZ_BUDGET_S=200 python scripts/fuzz_api_parity.py 300 98004 large 2>&1 | grep -v amdgpu | tail -3 >> $o
echo "# ... HYDAMD_RANS_WAVES=5" >> $o
HYDAMD_RANS_WAVES=5 python scripts/fuzz_api_parity.py 3000 98005 2>&1 | grep -v amdgpu | tail -3 >> $o
cat $o
