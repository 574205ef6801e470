#!/bin/bash
# the fuzz sweeps of a round, on the commit named in $1 (profiles/r05_fuzz.txt records it); ~25 min of one GPU
mkdir -p gpurun_out/r05fuzz; o=gpurun_out/r05fuzz/fuzz.txt
echo "# scripts/fuzz_api_parity.py on one MI355X, commit $1 (every case's codestream against the reference built from /root/reference, oracle/_ref)" > $o
python scripts/fuzz_api_parity.py 20000 50001 >> $o 2>&1
FUZZ_BUDGET_S=400 python scripts/fuzz_api_parity.py 600 50002 large >> $o 2>&1
echo "# HYDAMD_RANS_WAVES=5 (the drop-in API on the lane form)" >> $o
HYDAMD_RANS_WAVES=5 python scripts/fuzz_api_parity.py 6000 50003 >> $o 2>&1
HYDAMD_RANS_WAVES=5 FUZZ_BUDGET_S=200 python scripts/fuzz_api_parity.py 300 50004 large >> $o 2>&1
echo "# HYDAMD_TILE_PIPELINE=8" >> $o
HYDAMD_TILE_PIPELINE=8 python scripts/fuzz_api_parity.py 5000 50005 >> $o 2>&1
echo "# HYDAMD_DEVICES=0,0,0 HYDAMD_VERIFY_PEERS=1 (the in-library multi-device scheduler, aliased list, every peer view checked)" >> $o
HYDAMD_DEVICES=0,0,0 HYDAMD_VERIFY_PEERS=1 FUZZ_BUDGET_S=200 python scripts/fuzz_api_parity.py 300 50006 large >> $o 2>&1
echo "# HYDAMD_CURVE_GATHERS=2 (all six curves in registers)" >> $o
HYDAMD_CURVE_GATHERS=2 python scripts/fuzz_api_parity.py 4000 50007 >> $o 2>&1
cat $o
