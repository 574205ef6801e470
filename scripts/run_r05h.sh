mkdir -p gpurun_out/r05h; o=gpurun_out/r05h
python -m pytest tests/test_gpu_device_parity.py tests/test_gpu_fuzz.py tests/test_gpu_full_size.py tests/test_gpu_api_parity.py -m gpu -x -q > $o/tests.log 2>&1; grep -n "passed\|failed" $o/tests.log
for f in 4 5; do
  echo "== form $f" >> $o/one.log
  python scripts/one_frame.py 3 $f 2 t 2>&1 | grep -E "rans|bytes|rror" >> $o/one.log
done
cat $o/one.log
for i in 1 2; do
echo "== pipe new rans 5" >> $o/pipe.log; python scripts/pipe_probe.py --streams 16 --batch 2 --frames 512 --rans 5 --reps 2 >> $o/pipe.log 2>&1
echo "== pipe r04 rans 6" >> $o/pipe.log; HYDAMD_LIB=$PWD/scripts/probe_build/base_r04.so python scripts/pipe_probe.py --streams 16 --batch 2 --frames 512 --rans 6 --reps 2 >> $o/pipe.log 2>&1
echo "== pipe new rans 5 ride tables" >> $o/pipe.log; HYDAMD_LF_CODES_RIDE=tables python scripts/pipe_probe.py --streams 16 --batch 2 --frames 512 --rans 5 --reps 2 >> $o/pipe.log 2>&1
done
grep -E "==|SUSTAINED" $o/pipe.log
