#!/bin/bash
# Collect PMC counters for our kernels, one rocprofv3 --pmc pass per counter set (never combined
# with tracing).  usage: scripts/collect_pmc.sh <outdir under gpurun_out> <command...>
set -u
out=$1; shift
export TMPDIR=/tmp
mkdir -p "$out" /tmp/pmc
sets=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_ATOMIC"
 "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE SQ_WAVES"
 "FETCH_SIZE"
 "WRITE_SIZE"
)
i=0
for set in "${sets[@]}"; do
  i=$((i+1))
  rm -rf /tmp/pmc/p$i
  rocprofv3 --pmc $set --kernel-include-regex "k_transform|k_rans|k_build|k_pack|k_scan|k_lf|k_frame_begin|k_publish|k_asm|k_export" -d /tmp/pmc/p$i -o p -- "$@" > /tmp/pmc/log$i.txt 2>&1
  db=$(find /tmp/pmc/p$i -name "*.db" | head -1)
  if [ -n "$db" ]; then python scripts/pmc_summary.py "$db" > "$out/pmc_set$i.txt" 2>&1; else tail -5 /tmp/pmc/log$i.txt > "$out/pmc_set$i.txt"; fi
done
cat "$out"/pmc_set*.txt
