mkdir -p gpurun_out/r05l; o=gpurun_out/r05l
python -m pytest tests/test_gpu_device_parity.py tests/test_gpu_fuzz.py tests/test_gpu_api_parity.py tests/test_gpu_full_size.py -m gpu -x -q > $o/tests.log 2>&1; grep -n "passed\|failed\|Error" $o/tests.log | tail -5
for i in 1 2; do for l in "" scripts/probe_build/base_r04.so; do
  echo "== form 4 lib ${l:-new}" >> $o/one.log
  HYDAMD_LIB=${l:+$PWD/$l} python scripts/one_frame.py 3 4 2 t 2>&1 | grep -E "rans|rror" >> $o/one.log
done; done
cat $o/one.log
python scripts/api_tile_mode.py 4096 8 > $o/tile.log 2>&1; tail -5 $o/tile.log
HYDAMD_LIB=$PWD/scripts/probe_build/base_r04.so python scripts/api_tile_mode.py 4096 8 > $o/tile_old.log 2>&1; tail -5 $o/tile_old.log
