B=scripts/probe_build
for r in 1 2; do
for lib in nog old g08 tok15; do
  echo -n "$lib: "; HYDAMD_LIB=$PWD/$B/k1v_$lib.so GPU_MAX_HW_QUEUES=22 timeout 300 python scripts/pipe_probe.py --reps 1 --frames 768 --profile 0 --streams 32 --batch 2 2>&1 | grep SUSTAINED | cut -c60-120
done
done
