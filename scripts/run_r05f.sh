mkdir -p gpurun_out/r05f; o=gpurun_out/r05f
HYDAMD_CHAIN_CUS=32 python -m pytest tests/test_gpu_device_parity.py -m gpu -x -q > $o/tests_cu.log 2>&1; grep -n "passed\|failed" $o/tests_cu.log
echo "== baseline" >> $o/pipe.log; python scripts/pipe_probe.py --streams 16 --batch 2 --frames 512 --rans 5 --reps 2 >> $o/pipe.log 2>&1
for R in 16 24 32 40 48 64; do for cs in 4 6; do for r in 5 6; do for w in 1 2; do
  echo "== R $R chainstreams $cs rans $r waves $w" >> $o/pipe.log
  HYDAMD_LANES_WAVES=$w HYDAMD_CHAIN_CUS=$R HYDAMD_CHAIN_STREAMS=$cs timeout 120 python scripts/pipe_probe.py --streams 16 --batch 2 --frames 512 --rans $r --reps 1 >> $o/pipe.log 2>&1
done; done; done; done
echo "== baseline" >> $o/pipe.log; python scripts/pipe_probe.py --streams 16 --batch 2 --frames 512 --rans 5 --reps 2 >> $o/pipe.log 2>&1
grep -E "==|SUSTAINED|rror" $o/pipe.log
