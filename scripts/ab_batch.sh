#!/bin/bash
# A/B of two builds in --mode batch.  usage: bash scripts/ab_batch.sh <libA> <libB> [reps] [threads]
a=$1; b=$2; reps=${3:-3}; th=${4:-4}
for i in $(seq $reps); do
  for l in "$a" "$b"; do
    HYDAMD_LIB=$PWD/$l python bench.py --mode batch --threads $th 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$l', d['frames_per_s'], d['ms_per_step'])"
  done
done
