mkdir -p gpurun_out/r05n; o=gpurun_out/r05n
for i in 1 2; do for v in nc9 nc7 nc6 nc4; do
  echo "== $v" >> $o/pipe.log
  HYDAMD_LIB=$PWD/scripts/probe_build/k1v_$v.so python scripts/pipe_probe.py --streams 16 --batch 2 --frames 512 --rans 5 --reps 2 2>&1 | grep -E "SUSTAINED|rror" >> $o/pipe.log
done; done
for v in nc9 nc7 nc4; do echo "== $v alone" >> $o/pipe.log; HYDAMD_LIB=$PWD/scripts/probe_build/k1v_$v.so python scripts/one_frame.py 2 5 2 t 2>&1 | grep rans >> $o/pipe.log; done
cat $o/pipe.log
