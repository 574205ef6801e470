mkdir -p gpurun_out/r05p; o=gpurun_out/r05p
for i in 1 2; do for v in pr3 pr0 pr1; do
  echo "== $v" >> $o/pipe.log
  HYDAMD_LIB=$PWD/scripts/probe_build/k1v_$v.so python scripts/pipe_probe.py --streams 16 --batch 2 --frames 512 --rans 5 --reps 2 --profile 0 2>&1 | grep -E "SUSTAINED|rror" >> $o/pipe.log
done; done
cat $o/pipe.log
