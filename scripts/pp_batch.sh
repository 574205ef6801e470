pp() { GPU_MAX_HW_QUEUES=22 timeout 300 python scripts/pipe_probe.py --reps 1 --frames 512 --streams 16 --batch 2 --rans 6 "$@" 2>&1 | grep -E "SUSTAINED" | cut -c60-400; }
for us in 2500 5000; do for lds in 0 24000 45000 64000 92000; do
echo -n "sleep $us us, $lds B: "; HYDAMD_DEBUG_SKIP=16 HYDAMD_DEBUG_SLEEP_US=$us HYDAMD_DEBUG_SLEEP_LDS=$lds pp
done; done
echo -n "real chains: "; pp
