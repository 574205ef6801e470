mkdir -p gpurun_out/r05e; o=gpurun_out/r05e
for cfg in "1 32" "2 16" "3 16" "4 8" "4 16" "6 8" "8 4" "8 8" "12 4" "16 2" "16 4"; do set -- $cfg; for r in 5 6; do
  echo "== streams $1 batch $2 rans $r" >> $o/pipe.log
  fr=$(( 1024 / $2 )); timeout 300 python scripts/pipe_probe.py --streams $1 --batch $2 --frames $fr --rans $r --reps 2 >> $o/pipe.log 2>&1
done; done
grep -E "==|SUSTAINED|ms/frame" $o/pipe.log
