#!/bin/bash
# Everything under profiles/ for one round, in one go on the GPU box (copy gpurun_out/<tag>/<tag>_* to profiles/ afterwards):
#   the default bench line; kernel traces (rocprofv3 --kernel-trace --stats) of single frames in both entropy forms, of the
#   pipelined loop, of the 16K shard mode and of the drop-in API; the PMC passes (separate runs, never combined with tracing);
#   tile mode; the transform kernel by content and curve-gather choice; the removal table of the pipelined loop.
# Every probe's exit status and output are checked: a probe that fails leaves <name>.FAILED with its log's tail and the script
# exits non-zero at the end (round 4 committed a Python traceback as a profile).
# usage: bash scripts/collect_profiles.sh <tag>          e.g. r05
set -u
tag=$1
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
root=$PWD
failed=0

fail() { # name, log
  echo "PROBE FAILED: $1" >&2
  { echo "PROBE FAILED: $1"; tail -20 "$2" 2>/dev/null; } > "$out/${tag}_$1.FAILED"
  rm -f "$out/${tag}_$1.txt" "$out/${tag}_$1.json"
  failed=1
}
run() { # name, ext, command...  (stdout -> the profile, stderr -> a log; non-zero exit, a traceback or an empty file = failure)
  local name=$1 ext=$2; shift 2
  "$@" > "$out/${tag}_$name.$ext" 2> "/tmp/${tag}_$name.err"
  local rc=$?
  if [ $rc -ne 0 ] || [ ! -s "$out/${tag}_$name.$ext" ] || grep -qE "Traceback \(most recent call last\)|raise [A-Za-z]*Error|^[A-Za-z.]*Error: " "$out/${tag}_$name.$ext" "/tmp/${tag}_$name.err"; then
    cat "$out/${tag}_$name.$ext" >> "/tmp/${tag}_$name.err" 2>/dev/null
    fail "$name" "/tmp/${tag}_$name.err"
  fi
}
trace() { # name, summariser args..., -- command...: rocprofv3 kernel trace of the command, summarised into the profile
  local name=$1; shift
  local extra=()
  while [ "$1" != "--" ]; do extra+=("$1"); shift; done
  shift
  rm -rf "/tmp/kt_$name"
  (cd /tmp && rocprofv3 --kernel-trace "${extra[@]}" --stats -d "/tmp/kt_$name" -o kt -- "$@" > "/tmp/kt_$name.log" 2>&1)
  local rc=$?
  local db
  db=$(find "/tmp/kt_$name" -name "*.db" 2>/dev/null | head -1)
  if [ $rc -ne 0 ] || [ -z "$db" ]; then fail "kernel_stats_$name" "/tmp/kt_$name.log"; return; fi
  run "kernel_stats_$name" txt bash -c "python $root/scripts/rocpd_summary.py $db | grep -v 'at::native'"
}

run bench_default json python bench.py
trace single -- python "$root/scripts/one_frame.py" 5 5 2
trace single_form4 -- python "$root/scripts/one_frame.py" 5 4 2
trace pipe -- python "$root/bench.py" --steps 256 --no-cpu-baseline --no-api --no-legs
grep "^{" /tmp/kt_pipe.log | tail -1 > "$out/${tag}_bench_under_rocprof.json"; [ -s "$out/${tag}_bench_under_rocprof.json" ] || fail bench_under_rocprof /tmp/kt_pipe.log
trace shard -- python "$root/bench.py" --mode shard --steps 30
grep "^{" /tmp/kt_shard.log | tail -1 > "$out/${tag}_shard_under_rocprof.json"; [ -s "$out/${tag}_shard_under_rocprof.json" ] || fail shard_under_rocprof /tmp/kt_shard.log
trace api --memory-copy-trace -- python "$root/scripts/api_frame_times.py"
db=$(find /tmp/kt_api -name "*.db" 2>/dev/null | head -1)
if [ -n "$db" ]; then run api_timeline txt python scripts/rocpd_timeline.py "$db" 75; else fail api_timeline /tmp/kt_api.log; fi  # the last frame: uploads, kernels, read-back
run tile_mode txt bash -c 'echo "## default: every hyd_send_tile call ends with its tile frame (the reference s timing)"; python scripts/api_tile_mode.py 4096 8 | grep shift; echo "## eight tile frames in flight (HYDAMD_TILE_PIPELINE=8), GPU_MAX_HW_QUEUES=22"; GPU_MAX_HW_QUEUES=22 HYDAMD_TILE_PIPELINE=8 python scripts/api_tile_mode.py 4096 8 | grep shift'
run k1_content txt bash -c 'for g in 0 1 2; do echo "== HYDAMD_CURVE_GATHERS=$g (0 by the last frame, 1 always, 2 never)"; HYDAMD_CURVE_GATHERS=$g python scripts/k1_content.py; done'
# the removal table of the pipelined loop (HYDAMD_DEBUG_SKIP: 1 tables, 2 chains, 4 scan + emit, 8 LF coder; 16: sleeping wavefronts in the chains' place)
run pipeline_bounds txt bash -c '
  p() { python scripts/pipe_probe.py --streams 16 --batch 2 --frames 512 --rans 5 --reps 2 "$@" 2>&1 | grep -E "SUSTAINED|rror" | sed "s/.*: //" | tr "\n" " "; echo; }
  echo "# the pipelined loop (16 contexts x 2 frames per launch group, lane-form chains), sustained Gpixel/s, two runs each; one box"
  echo -n "whole frame:                          "; p
  echo -n "without scan + emit (skip 4):         "; HYDAMD_DEBUG_SKIP=4 p
  echo -n "without the chain kernel (skip 2):    "; HYDAMD_DEBUG_SKIP=2 p
  echo -n "without chains, scan, emit (skip 6):  "; HYDAMD_DEBUG_SKIP=6 p
  echo -n "without the LF coder (skip 8):        "; HYDAMD_DEBUG_SKIP=8 p
  t() { python scripts/pipe_probe.py --streams $1 --frames 512 --rans 5 --reps 2 --only-transform 2>&1 | grep -E "SUSTAINED|rror" | sed "s/.*: //" | tr "\n" " "; echo; }
  echo -n "transform kernel only, 16 contexts:   "; t 16
  echo -n "transform kernel only, 32 contexts:   "; t 32
  for us in 4000 1750; do for lds in 0 24576 45056 65536 81920; do
    echo -n "sleeping wavefronts for ${us} us holding ${lds} B of LDS in the chains place: "; HYDAMD_DEBUG_SKIP=16 HYDAMD_DEBUG_SLEEP_US=$us HYDAMD_DEBUG_SLEEP_LDS=$lds p
  done; done
  echo -n "LF code construction riding the table kernel (HYDAMD_LF_CODES_RIDE=tables): "; HYDAMD_LF_CODES_RIDE=tables p
'
bash scripts/collect_pmc.sh "$out/pmc" python scripts/one_frame.py 3 5 2 > /tmp/${tag}_pmc.log 2>&1
cat "$out"/pmc/pmc_set*.txt > "$out/${tag}_pmc_8k_photo.txt" 2>/dev/null
if ! grep -q "SQ_INSTS_VALU" "$out/${tag}_pmc_8k_photo.txt" || ! grep -q "FETCH_SIZE" "$out/${tag}_pmc_8k_photo.txt"; then fail pmc_8k_photo /tmp/${tag}_pmc.log; fi
ls -la "$out"
exit $failed
