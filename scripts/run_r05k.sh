mkdir -p gpurun_out/r05k; o=gpurun_out/r05k
for i in 1 2; do for l in "" scripts/probe_build/base_r04.so scripts/probe_build/varB_oldasm.so; do
  echo "== lib ${l:-new}" >> $o/api.log
  HYDAMD_LIB=${l:+$PWD/$l} python scripts/api_frame_times.py 2>&1 | tail -3 >> $o/api.log
done; done
cat $o/api.log
HYDAMD_TRACE=1 python scripts/api_frame_times.py 2>&1 | tail -60 > $o/trace_new.log
HYDAMD_TRACE=1 HYDAMD_LIB=$PWD/scripts/probe_build/base_r04.so python scripts/api_frame_times.py 2>&1 | tail -60 > $o/trace_old.log
