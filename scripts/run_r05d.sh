mkdir -p gpurun_out/r05d; o=gpurun_out/r05d
HYDAMD_LANES_WAVES=2 python -m pytest tests/test_gpu_device_parity.py tests/test_gpu_fuzz.py -m gpu -x -q > $o/tests_w2.log 2>&1; grep -n "passed\|failed" $o/tests_w2.log
for f in 5 6; do for w in 1 2; do
  echo "== form $f waves $w" >> $o/one.log
  HYDAMD_LANES_WAVES=$w python scripts/one_frame.py 3 $f 2 t 2>&1 | grep -E "rans|bytes|rror" >> $o/one.log
done; done
cat $o/one.log
for ride in chain tables; do for r in 5 6; do for w in 1 2; do echo "== pipe rans $r waves $w ride $ride" >> $o/pipe.log; HYDAMD_LF_CODES_RIDE=$ride HYDAMD_LANES_WAVES=$w python scripts/pipe_probe.py --streams 16 --batch 2 --frames 512 --rans $r --reps 2 >> $o/pipe.log 2>&1; done; done; done
grep -E "==|SUSTAINED" $o/pipe.log
