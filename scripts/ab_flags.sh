#!/bin/bash
# A/B of two sets of bench.py flags on one box, alternating.  usage: bash scripts/ab_flags.sh "<flags A>" "<flags B>" [reps]
a=$1; b=$2; reps=${3:-4}
for i in $(seq $reps); do
  for f in "$a" "$b"; do
    python bench.py --steps 120 --no-cpu-baseline --no-api $f 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$f]', d['value'], d['ms_per_step'])"
  done
done
