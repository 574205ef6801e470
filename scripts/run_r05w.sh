mkdir -p gpurun_out/r05w
python -m pytest tests/test_gpu_api_parity.py tests/test_gpu_device_parity.py -m gpu -x -q 2>&1 | tail -2
p() { python scripts/pipe_probe.py --streams 16 --batch 2 --frames 512 --rans 5 --reps 2 "$@" 2>&1 | grep -E "SUSTAINED|rror" | sed "s/.*: //" | tr "\n" " "; echo; }
{
echo "# the pipelined loop (16 contexts x 2 frames per launch group, lane-form chains), sustained Gpixel/s, two runs each; one box; final code"
echo -n "whole frame:                          "; p
echo -n "without scan + emit (skip 4):         "; HYDAMD_DEBUG_SKIP=4 p
echo -n "without the chain kernel (skip 2):    "; HYDAMD_DEBUG_SKIP=2 p
echo -n "without chains, scan, emit (skip 6):  "; HYDAMD_DEBUG_SKIP=6 p
echo -n "sleeping wavefronts 4000 us, 65536 B: "; HYDAMD_DEBUG_SKIP=16 HYDAMD_DEBUG_SLEEP_US=4000 HYDAMD_DEBUG_SLEEP_LDS=65536 p
} > gpurun_out/r05w/bounds_final.txt 2>&1
cat gpurun_out/r05w/bounds_final.txt
