mkdir -p gpurun_out/r05c; o=gpurun_out/r05c
for v in new p1 p2 p4 p5 p6 p8 p16; do
  echo "== variant $v" >> $o/one.log
  HYDAMD_LIB=$PWD/scripts/probe_build/k1v_$v.so python scripts/one_frame.py 2 5 2 t 2>&1 | grep -E "rans|Error|error" >> $o/one.log
  HYDAMD_LIB=$PWD/scripts/probe_build/k1v_$v.so python scripts/one_frame.py 2 6 2 t 2>&1 | grep -E "rans|Error|error" >> $o/one.log
done
cat $o/one.log
