#!/bin/bash
# A/B of two builds of the library on one GPU box, alternating runs so that clock and box drift hit both alike.
# usage: bash scripts/ab_bench.sh <libA> <libB> [reps] [extra bench.py args...]
a=$1; b=$2; reps=${3:-4}; shift 3
for i in $(seq $reps); do
  for l in "$a" "$b"; do
    HYDAMD_LIB=$PWD/$l python bench.py --steps 120 --no-cpu-baseline --no-api "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$l', d['value'], d['ms_per_step'], d['single_frame_form5']['ms_per_frame'], 'K1', d['single_frame_form5']['kernel_avg_ms']['transform_tokenize'])"
  done
done
