mkdir -p gpurun_out/r05g; o=gpurun_out/r05g
python -m pytest tests -m gpu -x -q > $o/tests.log 2>&1; grep -n "passed\|failed" $o/tests.log
for f in 4 5 6; do
  echo "== form $f" >> $o/one.log
  python scripts/one_frame.py 3 $f 2 t 2>&1 | grep -E "rans|bytes|rror" >> $o/one.log
done
cat $o/one.log
python scripts/api_frame_times.py > $o/api.log 2>&1; tail -5 $o/api.log
python scripts/api_tile_mode.py 4096 8 > $o/tile.log 2>&1; tail -12 $o/tile.log
