export TMPDIR=/tmp; root=$PWD; out=gpurun_out/r04; mkdir -p $out
cd /tmp; rm -rf /tmp/kt_single
rocprofv3 --kernel-trace --stats -d /tmp/kt_single -o kt -- python "$root/scripts/one_frame.py" 5 5 2 > /tmp/kt_single.log 2>&1
cd "$root"
db=$(find /tmp/kt_single -name "*.db" | head -1); python scripts/rocpd_summary.py "$db" | grep -v "at::native" > "$out/r04_kernel_stats_single.txt" 2>&1
rm -rf /tmp/kt_f4; cd /tmp; rocprofv3 --kernel-trace --stats -d /tmp/kt_f4 -o kt -- python "$root/scripts/one_frame.py" 5 4 1 > /tmp/kt_f4.log 2>&1; cd "$root"
db=$(find /tmp/kt_f4 -name "*.db" | head -1); python scripts/rocpd_summary.py "$db" | grep -v "at::native" > "$out/r04_kernel_stats_single_form4.txt" 2>&1
bash scripts/collect_pmc.sh "$out/pmc" python scripts/one_frame.py 3 5 2 > /dev/null 2>&1
cat "$out"/pmc/pmc_set*.txt > "$out/r04_pmc_8k_photo.txt"
head -12 $out/r04_kernel_stats_single.txt; head -8 $out/r04_kernel_stats_single_form4.txt
