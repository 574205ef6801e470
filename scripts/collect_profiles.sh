#!/bin/bash
# Everything under profiles/ for one round, in one go on the GPU box (copy gpurun_out/<tag>/<tag>_* to profiles/ afterwards):
#   the default bench line; kernel traces (rocprofv3 --kernel-trace --stats) of single frames in both entropy forms, of the
#   pipelined loop, of the 16K shard mode and of the drop-in API; the PMC passes (separate runs, never combined with tracing);
#   tile mode; the transform kernel by content and curve-gather choice; the removal table of the pipelined loop.
# Every probe's exit status and output are checked: a probe that fails leaves <name>.FAILED with its log's tail and the script
# exits non-zero at the end (round 4 committed a Python traceback as a profile).
# usage: bash scripts/collect_profiles.sh <tag> [target ...]      e.g. r06            (every target)
#                                                                     r06 chain_probes pmc
# One target per file under profiles/ (the name after the tag): bench_default, kernel_stats (single, single_form4, pipe, shard,
# api + api_timeline), tile_mode, k1_content, pipeline_bounds, chain_probes, priorities, lane_step, lane_pipe, loop_stage_times,
# emit_share, icache, pmc_8k_photo, fuzz, split_loop, gt_chain, noise_forms, nc_probe, pg_presence, stream_priorities, k1_waves5, chanseq, chain_lds_min, sleeper_matrix, dyn_lds (the two as first collected are several runs' outputs joined: profiles/r06_sleeper_matrix.txt, r06_dyn_lds.txt).  The kernel variants the probe targets load: bash scripts/build_probe_variants.sh (here, before gpurun).
# The probes that skip stages or run stand-in kernels load hydrium_amd/lib/libhydrium_probe.so (HYD_TEST_HOOKS flavour;
# scripts/pipe_probe.py selects it) or a variant built by `python scripts/k1_variants.py --build ...` (chain_probes and
# priorities build theirs HERE, before the gpurun call: hipcc cross-compiles, the .so files travel with the snapshot).
set -u
tag=$1; shift
targets=" ${*:-bench_default kernel_stats tile_mode k1_content pipeline_bounds chain_probes priorities lane_step lane_pipe loop_stage_times emit_share launch_boundaries icache pmc_8k_photo split_loop gt_chain noise_forms nc_probe pg_presence stream_priorities k1_waves5 chanseq chain_lds_min sleeper_matrix dyn_lds} "
want() { [[ "$targets" == *" $1 "* ]]; }
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

if want bench_default; then
run bench_default json python bench.py
fi

if want kernel_stats; then
trace single -- python "$root/scripts/one_frame.py" 5 5 2
trace single_form4 -- python "$root/scripts/one_frame.py" 5 4 2
trace pipe -- python "$root/bench.py" --steps 256 --no-cpu-baseline --no-api --no-legs
grep "^{" /tmp/kt_pipe.log | tail -1 > "$out/${tag}_bench_under_rocprof.json"; [ -s "$out/${tag}_bench_under_rocprof.json" ] || fail bench_under_rocprof /tmp/kt_pipe.log
# the same trace per hardware queue: how long every kernel of the loop WAITS behind its predecessor on its queue, and how long it runs
db=$(find /tmp/kt_pipe -name "*.db" 2>/dev/null | head -1)
if [ -n "$db" ]; then run stream_gaps txt bash -c "echo '# the pipelined loop under rocprofv3 --kernel-trace, per hardware queue: mean wait between the end of the previous kernel on the same queue and a kernel s start, and its mean duration (us); commit $(cat .commit 2>/dev/null)'; python scripts/rocpd_gaps.py $db 60 | grep -v '^columns\|^queues'"; else fail stream_gaps /tmp/kt_pipe.log; fi
trace shard -- python "$root/bench.py" --mode shard --steps 30
grep "^{" /tmp/kt_shard.log | tail -1 > "$out/${tag}_shard_under_rocprof.json"; [ -s "$out/${tag}_shard_under_rocprof.json" ] || fail shard_under_rocprof /tmp/kt_shard.log
trace api --memory-copy-trace -- python "$root/scripts/api_frame_times.py"
db=$(find /tmp/kt_api -name "*.db" 2>/dev/null | head -1)
if [ -n "$db" ]; then run api_timeline txt python scripts/rocpd_timeline.py "$db" 75; else fail api_timeline /tmp/kt_api.log; fi  # the last frame: uploads, kernels, read-back
fi

if want tile_mode; then
run tile_mode txt bash -c 'echo "## default: every hyd_send_tile call ends with its tile frame (the reference s timing)"; python scripts/api_tile_mode.py 4096 8 | grep shift; echo "## eight tile frames in flight (HYDAMD_TILE_PIPELINE=8), GPU_MAX_HW_QUEUES=22"; GPU_MAX_HW_QUEUES=22 HYDAMD_TILE_PIPELINE=8 python scripts/api_tile_mode.py 4096 8 | grep shift'
fi

if want k1_content; then
run k1_content txt bash -c 'for g in 0 1 2; do echo "== HYDAMD_CURVE_GATHERS=$g (0 by the last frame, 1 always, 2 never)"; HYDAMD_CURVE_GATHERS=$g python scripts/k1_content.py; done'
fi

# one sustained figure of the pipelined loop (16 contexts x 2 frames per launch group, lane-form chains), two repetitions
export PIPE_PROBE='p() { python scripts/pipe_probe.py --streams 16 --batch 2 --frames 512 --rans 5 --reps 2 "$@" 2>&1 | grep -E "SUSTAINED|rror" | sed "s/.*: //" | tr "\n" " "; echo; }'

# the removal table of the pipelined loop (HYDAMD_DEBUG_SKIP: 1 tables, 2 chains, 4 scan + emit, 8 LF coder; 16: stand-ins in
# the chains' place).  Every row without the real chain also skips scan + emit (their input is stale then and the emit kernel's
# consistency guard would stop it early): ONE commit, one box, comparable rows.
if want pipeline_bounds; then
run pipeline_bounds txt bash -c '
  eval "$PIPE_PROBE"
  echo "# the pipelined loop (16 contexts x 2 frames per launch group, lane-form chains), sustained Gpixel/s, two runs each; one box; commit $(cat .commit 2>/dev/null)"
  echo -n "whole frame:                                     "; p
  echo -n "without scan + emit (skip 4):                    "; HYDAMD_DEBUG_SKIP=4 p
  echo -n "without chains, scan, emit (skip 6):             "; HYDAMD_DEBUG_SKIP=6 p
  echo -n "without chains, scan, emit, LF coder (skip 14):  "; HYDAMD_DEBUG_SKIP=14 p
  echo -n "without the LF coder (skip 8):                   "; HYDAMD_DEBUG_SKIP=8 p
  t() { python scripts/pipe_probe.py --streams $1 --frames 512 --rans 5 --reps 2 --only-transform 2>&1 | grep -E "SUSTAINED|rror" | sed "s/.*: //" | tr "\n" " "; echo; }
  echo -n "transform kernel only, 16 contexts:              "; t 16
  for lds in 0 65536 81920; do for regs in 0 104; do
    echo -n "sleepers 2500 us, $regs VGPRs, $lds B LDS, no scan + emit (skip 20): "; HYDAMD_DEBUG_SKIP=20 HYDAMD_DEBUG_SLEEP_US=2500 HYDAMD_DEBUG_SLEEP_LDS=$lds HYDAMD_DEBUG_SLEEP_VGPRS=$regs p
  done; done
'
fi

# VERDICT r5 task 1: WHICH part of a chain's work costs the loop?  Stand-ins that do one part of it (device_api.hip
# k_chain_standin: 112 registers, 64 KB of LDS, 30 000 steps = the photo frame's longest group) and timing-only variants of
# the real chain (kernels.hip HYDK_CHAIN_PROBE), all with scan + emit off, on one box.
if want chain_probes; then
run chain_probes txt bash -c '
  eval "$PIPE_PROBE"
  v() { HYDAMD_LIB=$PWD/scripts/probe_build/k1v_$1.so; export HYDAMD_LIB; shift; "$@"; unset HYDAMD_LIB; }
  echo "# what a lane-form chain DOES, taken apart inside the pipelined loop; sustained Gpixel/s, two runs each; one box; commit $(cat .commit 2>/dev/null)"
  echo -n "real chain, no scan + emit (skip 4):                          "; HYDAMD_DEBUG_SKIP=4 p
  echo -n "no chain at all (skip 6):                                     "; HYDAMD_DEBUG_SKIP=6 p
  S="HYDAMD_DEBUG_SKIP=20 HYDAMD_DEBUG_SLEEP_LDS=65536"
  echo -n "sleepers, 104 VGPRs + 64 KB, 2500 us:                         "; env $S HYDAMD_DEBUG_SLEEP_VGPRS=104 HYDAMD_DEBUG_SLEEP_US=2500 bash -c "$PIPE_PROBE; p"
  echo -n "stand-in: the chain s 17 VALU instructions per step, no LDS:  "; env $S HYDAMD_DEBUG_STANDIN=valu bash -c "$PIPE_PROBE; p"
  echo -n "stand-in: its LDS reads only (random rows, conflicts):        "; env $S HYDAMD_DEBUG_STANDIN=lds bash -c "$PIPE_PROBE; p"
  echo -n "stand-in: both:                                               "; env $S HYDAMD_DEBUG_STANDIN=both bash -c "$PIPE_PROBE; p"
  echo -n "stand-in: the VALU load on FOUR wavefronts, a quarter each:   "; env $S HYDAMD_DEBUG_STANDIN=valu4 bash -c "$PIPE_PROBE; p"
  echo "#   control: equal durations (one wavefront: 18 200 steps last what four wavefronts need for 30 000 — 3.4 ms; four: 49 400 steps last what one needs for 30 000 — 5.6 ms)"
  echo -n "stand-in: VALU, one wavefront, 18200 steps (3.4 ms):          "; env $S HYDAMD_DEBUG_STANDIN=valu HYDAMD_DEBUG_STANDIN_STEPS=18200 bash -c "$PIPE_PROBE; p"
  echo -n "stand-in: VALU, four wavefronts, 49400 steps (5.6 ms):        "; env $S HYDAMD_DEBUG_STANDIN=valu4 HYDAMD_DEBUG_STANDIN_STEPS=49400 bash -c "$PIPE_PROBE; p"
  echo -n "stand-in: VALU, 80 KB of LDS:                                 "; env HYDAMD_DEBUG_SKIP=20 HYDAMD_DEBUG_SLEEP_LDS=81408 HYDAMD_DEBUG_STANDIN=valu bash -c "$PIPE_PROBE; p"
  echo -n "stand-in: VALU, no LDS held:                                  "; env HYDAMD_DEBUG_SKIP=20 HYDAMD_DEBUG_SLEEP_LDS=0 HYDAMD_DEBUG_STANDIN=valu bash -c "$PIPE_PROBE; p"
  for n in base q1 q4 q8 q16; do
    echo -n "real chain variant $n (HYDK_CHAIN_PROBE: q1 operand rows from one address, q4 no global traffic, q8 no stores, q16 no loads), skip 4: "; HYDAMD_DEBUG_SKIP=4 v $n p
  done
  echo "# the stand-ins and variants ALONE (one frame at a time, ms per chain launch by the library s event timers)"
  for k in valu lds both valu4; do echo -n "stand-in $k alone: "; HYDAMD_LIB=$PWD/hydrium_amd/lib/libhydrium_probe.so HYDAMD_DEBUG_SKIP=20 HYDAMD_DEBUG_SLEEP_LDS=65536 HYDAMD_DEBUG_STANDIN=$k python scripts/one_frame.py 2 5 2 t 2>&1 | grep rans_encode; done
  for n in base q1 q4 q8 q16; do echo -n "variant $n alone: "; HYDAMD_DEBUG_SKIP=4 v $n python scripts/one_frame.py 2 5 2 t 2>&1 | grep rans_encode; done
'
fi

# issue priorities: the transform kernel ABOVE the chains (rounds 4-5 only ever lowered the chains to the transform kernel s 0,
# where the oldest wavefront — the chain — still wins)
if want priorities; then
run priorities txt bash -c '
  eval "$PIPE_PROBE"
  v() { HYDAMD_LIB=$PWD/scripts/probe_build/k1v_$1.so; export HYDAMD_LIB; shift; "$@"; unset HYDAMD_LIB; }
  echo "# s_setprio of the transform kernel (k) and of the chain wavefronts (c); the product is k0 c3; whole frame, sustained Gpixel/s, alternating; commit $(cat .commit 2>/dev/null)"
  for rep in 1 2; do for n in base k1c0 k2c0 k3c0 k3c2; do echo -n "$n: "; v $n p; done; done
'
fi

# round 6's chain step against round 5's (kernels.hip HYDK_LANE_STEP): kernels alone, bytes, and the loop, alternating on one box
#   python scripts/k1_variants.py --build ls1=-DHYDK_LANE_STEP=1 ls2=-DHYDK_LANE_STEP=2     (before the gpurun call)
if want lane_step; then
run lane_step txt bash -c 'echo "# HYDK_LANE_STEP 1 (round 5: 13.5 vector instructions per symbol) against 2 (round 6: 11.5); commit $(cat .commit 2>/dev/null)"; python scripts/k1_variants.py --run --rounds 3 --pipe ls1 ls2'
fi

# how long the stages last INSIDE the loop (event timers around every stage of context 0: costs the loop ~4 %), real chain and variants
if want loop_stage_times; then
run loop_stage_times txt bash -c '
  echo "# stage durations inside the pipelined loop (scripts/pipe_probe.py --profile 1: library event timers, context 0), one box; commit $(cat .commit 2>/dev/null)"
  for n in r5 base q4; do
    echo "== chain variant $n (r5: round 5 s chain, HYDK_LANE_PIPE 0; base: the product s; q4: no global traffic, timing only)"
    HYDAMD_LIB=$PWD/scripts/probe_build/k1v_$n.so python scripts/pipe_probe.py --streams 16 --batch 2 --frames 256 --rans 5 --reps 2 --profile 1 2>&1 | grep -E "SUSTAINED|stage times|rror"
  done
  echo "== when do the chain wavefronts of a launch start and end? (HYDK_CHAIN_PROBE 32, scan + emit off; last line: alone)"
  for i in 1 2 3; do HYDAMD_DEBUG_SKIP=4 HYDAMD_LIB=$PWD/scripts/probe_build/k1v_q32.so python scripts/pipe_probe.py --streams 16 --batch 2 --frames 256 --rans 5 --reps 1 --chain-clock --profile 1 2>&1 | grep -E "SUSTAINED|chain wavefronts|stage times|rror"; done
  HYDAMD_DEBUG_SKIP=4 HYDAMD_LIB=$PWD/scripts/probe_build/k1v_q32.so python scripts/pipe_probe.py --streams 1 --batch 2 --frames 8 --rans 5 --reps 1 --chain-clock 2>&1 | grep -E "chain wavefronts|rror"
  echo "== a chain wavefront that owns its SIMD (HYDK_CHAIN_HOG)"
  for n in base hog; do echo -n "$n: "; HYDAMD_LIB=$PWD/scripts/probe_build/k1v_$n.so python scripts/pipe_probe.py --streams 16 --batch 2 --frames 384 --rans 5 --reps 2 --profile 1 2>&1 | grep -E "SUSTAINED|stage times|rror" | sed "s/^ *//" | tr "\n" " "; echo; done
  echo "== product library, alone (one frame at a time)"; python scripts/one_frame.py 2 5 2 t 2>&1 | grep -E "transform|rans|pack|tables|lf_"
'
fi

# how the chain's record lines travel (kernels.hip HYDK_LANE_PIPE 0 = round 5, 1 = two buffers, 2 = two pairs) and the step (HYDK_LANE_STEP)
#   python scripts/k1_variants.py --build r5=-DHYDK_LANE_PIPE=0,-DHYDK_LANE_STEP=1 pp1=-DHYDK_LANE_PIPE=1 pp2=-DHYDK_LANE_PIPE=2 pp2s2=-DHYDK_LANE_PIPE=2,-DHYDK_LANE_STEP=2
if want lane_pipe; then
run lane_pipe txt bash -c 'echo "# the lane-form chain: round 5 (r5) against two buffers taking turns (pp1), two pairs with the lines two rounds ahead (pp2; pp2s2: with the 11.5-instruction step); kernels alone, bytes, the photo loop and the noise loop, alternating on one box; commit $(cat .commit 2>/dev/null)"; K1V_NOISE=1 python scripts/k1_variants.py --run --rounds 3 --pipe r5 pp1 pp2 pp2s2'
fi

# transform workgroups that last a quarter / half as long in the pipelined loop (HYDAMD_K1_SPLIT_SLOTS=32: every launch of the loop is
# split as the one- and two-LF-group launches are, k_join_parts behind it): does a chain workgroup find its 80 KB sooner?
if want split_loop; then
run split_loop txt bash -c '
  eval "$PIPE_PROBE"
  echo "# HYDAMD_K1_SPLIT_SLOTS=32 (every transform launch of the loop split over 4 or 2 workgroups per group + k_join_parts), the pipelined loop, sustained Gpixel/s, alternating; commit $(cat .commit 2>/dev/null)"
  for i in 1 2 3; do
    echo -n "default:            "; p
    echo -n "split 32, 4 parts:  "; HYDAMD_K1_SPLIT_SLOTS=32 p
    echo -n "split 32, 2 parts:  "; HYDAMD_K1_SPLIT_SLOTS=32 HYDAMD_K1_SPLIT_LOG=1 p
  done
'
fi

# the lane-form chain with its slot tables read where the table kernel leaves them (L2) instead of a copy in LDS (kernels.hip
# HYDK_LANE_TAB_GLOBAL: 6 KB of LDS per chain workgroup instead of 80)
#   python scripts/k1_variants.py --build base= gt=-DHYDK_LANE_TAB_GLOBAL=1 gt1=-DHYDK_LANE_TAB_GLOBAL=1,-DHYDK_LANE_PIPE=1
if want gt_chain; then
run gt_chain txt bash -c '
  V=$PWD/scripts/probe_build
  pp() { python scripts/pipe_probe.py --frames 512 --rans 5 --reps 2 "$@" 2>&1 | grep -E "SUSTAINED|stage times|rror" | sed "s/.*: //" | tr "\n" " "; echo; }
  echo "# the lane-form chain with slot tables in L2 (gt; gt1: with HYDK_LANE_PIPE 1) against the product chain (base): kernels alone, bytes, the loop; commit $(cat .commit 2>/dev/null)"
  python scripts/k1_variants.py --run --rounds 2 --pipe base gt gt1
  echo "# the loop by frames per launch group and streams (scripts/pipe_probe.py, sustained Gpixel/s)"
  for b in 2 4 8; do for s in 16 22; do
    echo -n "base streams $s batch $b: "; HYDAMD_LIB=$V/k1v_base.so pp --streams $s --batch $b
    echo -n "gt   streams $s batch $b: "; HYDAMD_LIB=$V/k1v_gt.so pp --streams $s --batch $b
  done; done
  echo "# stage times in the loop (library event timers, context 0), four frames per launch group"
  for n in base gt; do echo "$n:"; HYDAMD_LIB=$V/k1v_$n.so python scripts/pipe_probe.py --streams 16 --batch 4 --frames 256 --rans 5 --reps 1 --profile 1 2>&1 | grep -E "SUSTAINED|stage times|rror"; done
  echo "# the same with the transform kernel s curve gather off (HYDAMD_CURVE_GATHERS=2: the texture path left to the chains)"
  export HYDAMD_CURVE_GATHERS=2
  for b in 2 4; do
    echo -n "base streams 16 batch $b: "; HYDAMD_LIB=$V/k1v_base.so pp --streams 16 --batch $b
    echo -n "gt   streams 16 batch $b: "; HYDAMD_LIB=$V/k1v_gt.so pp --streams 16 --batch $b
  done
'
fi

# the noise loop by entropy-stage form (VERDICT r5 task 7: would the wave form, faster alone on long groups, serve a chain-bound loop?)
if want noise_forms; then
run noise_forms txt bash -c '
  n() { python bench.py --kind noise --steps 48 --no-cpu-baseline --no-api --no-legs --no-content "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d[\"value\"]/1e3,1), \"Gpixel/s\", d[\"timing\"][\"Mpixel/s_each_window\"])"; }
  echo "# the noise loop (8192^2 RGB16 random pixels, 2.86 symbols per pixel) by entropy-stage form: 5 = one lane per group (79.5 KB of LDS per LF group), 4 = one wave per group (144 KB per workgroup of four groups); bench.py --kind noise; commit $(cat .commit 2>/dev/null)"
  for i in 1 2; do
    echo -n "lane form, 16 streams x 2: "; n --rans-waves 5
    echo -n "wave form, 16 streams x 2: "; n --rans-waves 4
    echo -n "wave form, 16 streams x 1: "; n --rans-waves 4 --frames-per-launch 1
    echo -n "wave form,  8 streams x 2: "; n --rans-waves 4 --streams 8
  done
'
fi

# timing only: what would a chain holding fewer kilobytes be worth (kernels.hip HYDK_LANE_NC9_PROBE)?
#   python scripts/k1_variants.py --build base= pp1=-DHYDK_LANE_PIPE=1 pp1n7=-DHYDK_LANE_PIPE=1,-DHYDK_LANE_NC9_PROBE=7 pp1n4=-DHYDK_LANE_PIPE=1,-DHYDK_LANE_NC9_PROBE=4 pp2n7=-DHYDK_LANE_NC9_PROBE=7 pp2n4=-DHYDK_LANE_NC9_PROBE=4
if want nc_probe; then
run nc_probe txt bash -c '
  eval "$PIPE_PROBE"
  v() { HYDAMD_LIB=$PWD/scripts/probe_build/k1v_$1.so; export HYDAMD_LIB; shift; "$@"; unset HYDAMD_LIB; }
  echo "# timing only: a nine-cluster frame s chains holding tables for 7 / 4 clusters (61.8 / 35 KB instead of 79.5), with the chain at 128 registers (HYDK_LANE_PIPE 1: three transform wavefronts fit beside it on its SIMD) and at 164 (PIPE 2: two); the pipelined loop without scan + emit (HYDAMD_DEBUG_SKIP=4), sustained Gpixel/s; commit $(cat .commit 2>/dev/null)"
  for i in 1 2; do for n in base pp1 pp1n7 pp1n4 pp2n7 pp2n4; do echo -n "$n (skip 4): "; HYDAMD_DEBUG_SKIP=4 v $n p; done; done
  echo "# the chain kernels alone"
  python scripts/k1_variants.py --run --rounds 1 base pp1 pp1n7 pp1n4 pp2n7 pp2n4 | grep -v "^$"
'
fi

# what an RCCL process group's mere presence in the process costs the frame loop (every rank of a --gpus N job carries one)
if want pg_presence; then
run pg_presence txt bash -c '
  b() { python bench.py --steps 256 --no-cpu-baseline --no-api --no-legs --no-content "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d[\"value\"]/1e3,1), \"Gpixel/s\", d[\"timing\"][\"Mpixel/s_each_window\"], \"rccl_ranks\", d.get(\"rccl_ranks\"))"; }
  echo "# the frame loop on one GPU without and with an RCCL process group in the process (HYDAMD_BENCH_FORCE_PG=1: what every rank of a --gpus N job carries); commit $(cat .commit 2>/dev/null)"
  for i in 1 2 3; do echo -n "no process group:   "; b; echo -n "RCCL group present: "; HYDAMD_BENCH_FORCE_PG=1 b; done
'
fi

# a staggered mix of stages: some of the contexts' streams at the device's highest priority (HYDAMD_STREAM_HIGH)
if want stream_priorities; then
run stream_priorities txt bash -c '
  eval "$PIPE_PROBE"
  echo "# the pipelined loop with the first n of the sixteen contexts own streams created at the device s highest priority (HYDAMD_STREAM_HIGH=n), sustained Gpixel/s; commit $(cat .commit 2>/dev/null)"
  for i in 1 2; do for n in 0 2 4 8 16; do echo -n "high $n of 16:    "; HYDAMD_STREAM_HIGH=$n p; done; done
'
fi

# the transform kernel compiled for five wavefronts per SIMD (python scripts/k1_variants.py --build base= w5=-DHYDK_K1_WAVES=5 w5i1=-DHYDK_K1_WAVES=5,-DHYDK_K1_ILP=1 i1=-DHYDK_K1_ILP=1)
if want k1_waves5; then
run k1_waves5 txt bash -c 'echo "# the transform kernel compiled for five wavefronts per SIMD (96 registers, 14 / 7 spilled) with two pixels / one pixel in lock step; alone, bytes, the bench loop; commit $(cat .commit 2>/dev/null)"; python scripts/k1_variants.py --run --rounds 2 --pipe base w5 w5i1 i1 | grep -v "^$"'
fi

# the transform kernel at 21 LDS granules instead of 25 (kernels.hip HYDK_K1_CHANSEQ): python scripts/k1_variants.py --build base= pp1=-DHYDK_LANE_PIPE=1 cs=-DHYDK_K1_CHANSEQ=1 cs1=-DHYDK_K1_CHANSEQ=1,-DHYDK_LANE_PIPE=1
if want chanseq; then
run chanseq txt bash -c '
  eval "$PIPE_PROBE"
  v() { HYDAMD_LIB=$PWD/scripts/probe_build/k1v_$1.so; export HYDAMD_LIB; shift; "$@"; unset HYDAMD_LIB; }
  t() { python scripts/pipe_probe.py --streams 16 --frames 512 --rans 5 --reps 2 --only-transform 2>&1 | grep -E "SUSTAINED|rror" | sed "s/.*: //" | tr "\n" " "; echo; }
  echo "# HYDK_K1_CHANSEQ (cs: the channels take the row-pass buffer in turn, 16-bit quantised coefficients: 25.7 KB of LDS per transform workgroup instead of 31.9; cs1: with the chain at 128 registers, HYDK_LANE_PIPE 1, so that three transform workgroups fit beside a chain); alone, bytes, the bench loop; commit $(cat .commit 2>/dev/null)"
  python scripts/k1_variants.py --run --rounds 2 --pipe base pp1 cs cs1 | grep -v "^$"
  echo "# transform kernels only, back to back on 16 streams (capacity), and the loop without scan + emit"
  for i in 1 2; do
    for n in base cs; do echo -n "$n transform only: "; v $n t; done
    for n in base pp1 cs cs1; do echo -n "$n (skip 4): "; HYDAMD_DEBUG_SKIP=4 v $n p; done
  done
'
fi

# a chain workgroup that asks for more than half a compute unit's LDS (kernels.hip HYDK_CHAIN_LDS_MIN; python scripts/k1_variants.py --build base= pad65=-DHYDK_CHAIN_LDS_MIN=83200 pad70=-DHYDK_CHAIN_LDS_MIN=89600 pad78=-DHYDK_CHAIN_LDS_MIN=99840)
if want chain_lds_min; then
run chain_lds_min txt bash -c 'echo "# HYDK_CHAIN_LDS_MIN: a chain workgroup asks for 65 / 70 / 78 LDS granules instead of 63, so that a compute unit holds ONE chain at most (two chains of 63 leave no room for a transform workgroup); alone, bytes, the bench loop; commit $(cat .commit 2>/dev/null)"; python scripts/k1_variants.py --run --rounds 3 --pipe base pad65 pad70 pad78 | grep -v "^$"'
fi

# sleepers in the chains' place: LDS held x registers held x duration, beside the 25-granule transform kernel (base) and the 21-granule one (cs)
if want sleeper_matrix; then
run sleeper_matrix txt bash -c '
  eval "$PIPE_PROBE"
  v() { HYDAMD_LIB=$PWD/scripts/probe_build/k1v_$1.so; export HYDAMD_LIB; shift; "$@"; unset HYDAMD_LIB; }
  echo "# sleeping wavefronts in the chains place (HYDAMD_DEBUG_SKIP=20: no chains, no scan + emit): bytes of LDS x registers x microseconds; sustained Gpixel/s; commit $(cat .commit 2>/dev/null)"
  for n in base cs; do
    echo -n "$n real chain (skip 4): "; HYDAMD_DEBUG_SKIP=4 v $n p
    echo -n "$n no chain (skip 6):   "; HYDAMD_DEBUG_SKIP=6 v $n p
    for us in 1800 4000; do for lds in 0 32768 62720 65536 72960 80640 83200 99840; do
      echo -n "$n $us us, 104 registers, $lds B: "; HYDAMD_DEBUG_SKIP=20 HYDAMD_DEBUG_SLEEP_LDS=$lds HYDAMD_DEBUG_SLEEP_VGPRS=104 HYDAMD_DEBUG_SLEEP_US=$us v $n p
    done; done
    for r in 104 124 160; do echo -n "$n 1800 us, $r registers, 65536 B: "; HYDAMD_DEBUG_SKIP=20 HYDAMD_DEBUG_SLEEP_LDS=65536 HYDAMD_DEBUG_SLEEP_VGPRS=$r HYDAMD_DEBUG_SLEEP_US=1800 v $n p; done
    for r in 104 124 160; do echo -n "$n 1800 us, $r registers, 80640 B: "; HYDAMD_DEBUG_SKIP=20 HYDAMD_DEBUG_SLEEP_LDS=80640 HYDAMD_DEBUG_SLEEP_VGPRS=$r HYDAMD_DEBUG_SLEEP_US=1800 v $n p; done
  done
'
fi

# the chain at its true register allocation (HYDK_CHAIN_DYN_LDS) beside the 21-granule transform kernel, and what the LF coder costs there
if want dyn_lds; then
run dyn_lds txt bash -c '
  eval "$PIPE_PROBE"
  v() { HYDAMD_LIB=$PWD/scripts/probe_build/k1v_$1.so; export HYDAMD_LIB; shift; "$@"; unset HYDAMD_LIB; }
  echo "# dyn: the chain s LDS asked for at launch (164 registers allocated instead of 264); csdynp1: + HYDK_K1_CHANSEQ + HYDK_LANE_PIPE 1 (128); commit $(cat .commit 2>/dev/null)"
  K1V_NOISE=1 python scripts/k1_variants.py --run --rounds 2 --pipe base dyn csdynp1 | grep -v "^$"
  echo "# loop without scan + emit (skip 4), with and without the LF coder; q5: the chain without global traffic and bank conflicts (timing only)"
  for n in base csdynp1; do echo -n "$n (skip 4): "; HYDAMD_DEBUG_SKIP=4 v $n p; echo -n "$n (skip 4), LF coder off: "; HYDAMD_DEBUG_SKIP=4 v $n p --lf 0; done
  echo -n "csdynp1q5 (skip 4), LF coder off: "; HYDAMD_DEBUG_SKIP=4 v csdynp1q5 p --lf 0
  echo "# the full loop; LF kernels dropped (HYDAMD_DEBUG_LF_DROP: 1 tokens, 2 code passengers, 4 offsets + pack, 8 gather) and timing-only token kernels (HYDK_LF_PROBE: 2 no histograms, 4 no records, 8 workgroups return at once)"
  for n in base csdynp1; do echo -n "$n: "; v $n p; echo -n "$n, LF coder off: "; v $n p --lf 0; done
  for d in 1 2 4 8 15; do echo -n "csdynp1 drop $d: "; HYDAMD_DEBUG_LF_DROP=$d v csdynp1 p; done
  for n in lp2 lp4 lp8; do echo -n "$n: "; v $n p; done
'
fi

# the emit kernel as fewer, fatter workgroups (HYDAMD_EMIT_SHARE virtual blocks per workgroup): does a small kernel wait for its
# workgroups turns among the transform kernels of fifteen other queues?
if want emit_share; then
run emit_share txt bash -c '
  eval "$PIPE_PROBE"
  echo "# HYDAMD_EMIT_SHARE (virtual blocks of four groups per emit workgroup), the pipelined loop, sustained Gpixel/s, alternating; commit $(cat .commit 2>/dev/null)"
  for rep in 1 2; do for sh in 1 2 4 8; do echo -n "share $sh: "; HYDAMD_EMIT_SHARE=$sh p; done; done
'
fi

# what does a kernel BOUNDARY cost a stream in the loop (empty kernels appended to every launch group), do the runtime's
# system-scope event releases matter, and one hardware queue's timeline kernel by kernel
if want launch_boundaries; then
run launch_boundaries txt bash -c '
  eval "$PIPE_PROBE"
  echo "# the pipelined loop, sustained Gpixel/s; commit $(cat .commit 2>/dev/null)"
  for rep in 1 2; do for n in 0 2 4 8 16; do echo -n "extra empty single-wavefront kernels per launch group (HYDAMD_DEBUG_EXTRA_LAUNCHES) $n: "; HYDAMD_DEBUG_EXTRA_LAUNCHES=$n p; done; done
  for rep in 1 2; do for sc in system device; do for ev in torch device; do echo -n "order-only events of the library release at $sc scope, the probe s own events: $ev: "; HYDAMD_EVENT_SCOPE=$sc p --events $ev; done; done; done
  rm -rf /tmp/kt_q; (cd /tmp && rocprofv3 --kernel-trace -d /tmp/kt_q -o kt -- python '"$root"'/scripts/pipe_probe.py --streams 16 --batch 2 --frames 256 --rans 5 --reps 1 > /tmp/kt_q.log 2>&1)
  python scripts/rocpd_queue_timeline.py $(find /tmp/kt_q -name "*.db" | head -1) 36 3
'
fi

# instruction cache: the transform kernel is 29.6 KB of code, the chain kernel 17 KB, the table kernel 23 KB (llvm-readelf -s)
if want icache; then
mkdir -p /tmp/pmc_ic
(cd /tmp && rocprofv3 -L > /tmp/pmc_ic/list.txt 2>&1; grep -i -E "ICACHE|IFETCH|INST_CACHE" /tmp/pmc_ic/list.txt | sort -u | head -40 > "$root/$out/${tag}_icache_counters_available.txt")
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  n=$(echo $set | cut -d" " -f1)
  rm -rf /tmp/pmc_ic/$n
  (cd /tmp && rocprofv3 --pmc $set --kernel-include-regex "k_transform|k_rans" -d /tmp/pmc_ic/$n -o p -- python "$root/scripts/pipe_probe.py" --streams 16 --batch 2 --frames 64 --rans 5 --reps 1 > /tmp/pmc_ic/$n.log 2>&1)
  db=$(find /tmp/pmc_ic/$n -name "*.db" | head -1)
  if [ -n "$db" ]; then python scripts/pmc_summary.py "$db" > "$out/${tag}_icache_loop_$n.txt" 2>&1; else tail -5 /tmp/pmc_ic/$n.log > "$out/${tag}_icache_loop_$n.FAILED"; fi
  rm -rf /tmp/pmc_ic/${n}_alone
  (cd /tmp && rocprofv3 --pmc $set --kernel-include-regex "k_transform|k_rans" -d /tmp/pmc_ic/${n}_alone -o p -- python "$root/scripts/one_frame.py" 3 5 2 > /tmp/pmc_ic/${n}_alone.log 2>&1)
  db=$(find /tmp/pmc_ic/${n}_alone -name "*.db" | head -1)
  if [ -n "$db" ]; then python scripts/pmc_summary.py "$db" > "$out/${tag}_icache_alone_$n.txt" 2>&1; fi
done
fi

if want pmc_8k_photo; then
bash scripts/collect_pmc.sh "$out/pmc" python scripts/one_frame.py 3 5 2 > /tmp/${tag}_pmc.log 2>&1
cat "$out"/pmc/pmc_set*.txt > "$out/${tag}_pmc_8k_photo.txt" 2>/dev/null
if ! grep -q "SQ_INSTS_VALU" "$out/${tag}_pmc_8k_photo.txt" || ! grep -q "FETCH_SIZE" "$out/${tag}_pmc_8k_photo.txt"; then fail pmc_8k_photo /tmp/${tag}_pmc.log; fi
fi

# seeded fuzz of the drop-in API against the reference (scripts/fuzz_api_parity.py); not part of the default set
if want fuzz; then
run fuzz txt bash -c 'echo "# commit $(cat .commit 2>/dev/null)"; for seed in 61 62 63; do FUZZ_BUDGET_S=${FUZZ_BUDGET_S:-600} python scripts/fuzz_api_parity.py 4000 $seed 2>&1 | tail -3; done; FUZZ_BUDGET_S=${FUZZ_BUDGET_S:-600} python scripts/fuzz_api_parity.py 400 64 large 2>&1 | tail -3; echo "# large cases with the device list aliased (HYDAMD_DEVICES=0,0,0: frames of 8 and more LF groups are dealt to three contexts, the home entry rotating, peer reads verified at first use)"; HYDAMD_DEVICES=0,0,0 FUZZ_BUDGET_S=${FUZZ_BUDGET_S:-600} python scripts/fuzz_api_parity.py 400 65 large 2>&1 | tail -3'
fi
ls -la "$out"
exit $failed
