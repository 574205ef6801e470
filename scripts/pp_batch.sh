for cfg in "24 4 20" "24 4 16" "24 6 18" "24 3 21" "26 4 22" "28 4 24" "22 4 18" "24 8 16"; do
  set -- $cfg
  echo -n "split, idle streams dropped: queues $1 lanes $2 contexts $3: "; GPU_MAX_HW_QUEUES=$1 timeout 300 python scripts/pipe_probe.py --reps 2 --frames 256 --profile 0 --split 1 --lanes $2 --streams $3 2>&1 | tail -2 | cut -c40-83 | tr '\n' ' '; echo
done
