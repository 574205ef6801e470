mkdir -p gpurun_out/r05b; o=gpurun_out/r05b
python -m pytest tests/test_gpu_device_parity.py tests/test_gpu_fuzz.py tests/test_gpu_full_size.py -m gpu -x -q > $o/tests.log 2>&1; grep -n "passed\|failed" $o/tests.log
for f in 5 6 4; do
 for l in "" scripts/probe_build/base_r04.so; do
  echo "== form $f lib ${l:-new}" >> $o/one.log
  HYDAMD_LIB=${l:+$PWD/$l} python scripts/one_frame.py 3 $f 2 t >> $o/one.log 2>&1
 done
done
grep -E "==|rans|section" $o/one.log
for r in 5 6; do for l in "" scripts/probe_build/base_r04.so; do echo "== pipe rans $r lib ${l:-new}" >> $o/pipe.log; HYDAMD_LIB=${l:+$PWD/$l} python scripts/pipe_probe.py --streams 16 --batch 2 --frames 512 --rans $r --reps 2 >> $o/pipe.log 2>&1; done; done
grep -E "==|SUSTAINED" $o/pipe.log
