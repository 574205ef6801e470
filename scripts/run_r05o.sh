mkdir -p gpurun_out/r05o; o=gpurun_out/r05o
p() { python scripts/pipe_probe.py --streams 16 --batch 2 --frames 512 --rans 5 --reps 2 2>&1 | grep -E "SUSTAINED|rror" | sed "s/.*: //" | tr "\n" " "; echo; }
{
echo -n "whole frame: "; p
echo -n "without the chain kernel: "; HYDAMD_DEBUG_SKIP=2 p
for regs in 0 40 104; do for lds in 0 32768 65536; do
  echo -n "sleepers 4000 us, $regs VGPRs, $lds B LDS: "; HYDAMD_DEBUG_SKIP=16 HYDAMD_DEBUG_SLEEP_US=4000 HYDAMD_DEBUG_SLEEP_LDS=$lds HYDAMD_DEBUG_SLEEP_VGPRS=$regs p
done; done
} > $o/sleep.log 2>&1
cat $o/sleep.log
