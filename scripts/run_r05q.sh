mkdir -p gpurun_out/r05q; o=gpurun_out/r05q
for i in 1 2; do for v in w4 w5 w5i1; do
  echo "== $v" >> $o/log
  HYDAMD_LIB=$PWD/scripts/probe_build/k1v_$v.so python scripts/one_frame.py 3 5 2 t 2>&1 | grep -E "transform|rror" >> $o/log
  HYDAMD_LIB=$PWD/scripts/probe_build/k1v_$v.so python scripts/pipe_probe.py --streams 16 --batch 2 --frames 512 --rans 5 --reps 2 2>&1 | grep -E "SUSTAINED|rror" >> $o/log
done; done
cat $o/log
