#!/usr/bin/env python3
"""bench.py — Mpixel/s of hydrium's per-group encode hot path on MI355X (driver contract).

Workload (BASELINE.json configs[2], the one the metric is quoted on): one 8192x8192 RGB16 frame
per GPU — 16 LF groups, 1024 groups of 256x256 — of the deterministic "photo" content
(SURVEY.md Appendix C), already resident in HBM when the timed region starts.  One "step" is one
pass of the whole hot path over that frame: RGB->XYB, 8x8 DCT, quantisation, tokenisation +
histograms (k_transform_tokenize), ANS tables (k_build_tables), rANS coding of all 1024 group
sections (k_rans_encode) and their packing (k_scan_sections / k_pack_sections), through the
additive C-ABI of include/hydrium_amd.h.  Consecutive steps are spread over `--streams`
independent contexts so that the latency-bound rANS kernel of one frame overlaps the transform
kernel of the next, as a production encoder serving a queue of frames would.

N > 1: weak scaling — one process per GPU over RCCL, rank r codes its own 8192x8192 frames (windows of one larger
synthetic image); no collective in the data path (the frames are independent), only the barriers and the clock are
shared (`--exchange` adds a per-frame blob gather to a rotating rank).  `python bench.py --gpus N` starts the N ranks
ITSELF (re-executing under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N`) when it is not already
running inside a process group; launched by torch.distributed.run it checks that the world it finds is the `--gpus` it
was given.  `n_gpus` is the process group's size and `rccl_ranks` the sum of an all-reduce of ones over it.
`--mode shard` is the strong-scaling leg (ONE 16384x16384 frame per step, LF groups sharded over the ranks, one blob
gather per frame to the assembling rank).  The frame-mode line carries the other workloads as legs: on one GPU they run
in-process behind the timed loop; at N > 1 the legs that need every rank (shard_16k, batch_4k) are started by rank 0 as jobs of
their own once the frame loop's process group is gone (`child_job`: `bench.py --gpus N --mode ...`, 300 s, killed as a whole on
overrun), so that a leg that fails or hangs cannot take the headline with it.

Prints ONE JSON line on rank 0.  `value` is whole-job Mpixel/s over all GPUs.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# Frames in flight live on HIP streams of their own, which the runtime deals to GPU_MAX_HW_QUEUES hardware queues in
# rotation, every stream the process creates taking a turn whether it carries work or not.  Until a context's LF side
# stream became lazy (device_api.hip ensure_lf_stream) the contexts' main streams sat on every second queue and the best
# shape was an accident of the rotation (32 contexts on 22 queues, 140 Gpixel/s; 24 on 22: 118).  With one stream per
# context the loop is flat: 12-20 contexts, each on a queue of its own (20-28 queues), two frames per launch group:
# 145-149 Gpixel/s sustained (profiles/r04_queue_stream_scan.txt).  MORE THAN 22 QUEUES IN USE and the firmware
# time-slices them: the host-pointer batch leg (8-10 encoder threads, each context a kernel stream and an LF side stream,
# plus the shared upload streams) runs 1400 frames/s on 20-22 queues and 500 on 24-28.  So: 22 queues, 16 contexts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "22")

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def emit(out):
    """Print the result line and make it the LAST thing on stdout: librccl writes a version banner to
    stdout when the process exits, after anything Python can print."""
    print(json.dumps(out), flush=True)
    sys.stdout.flush()
    os.dup2(os.open(os.devnull, os.O_WRONLY), 1)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="GPUs of this node to run on, one rank each (default: the world size torch.distributed.run gave us, else 1); "
                         "with N > 1 and no process group in the environment bench.py launches the N ranks itself")
    ap.add_argument("--dry-run-launch", action="store_true",
                    help="start the ranks exactly as a real run does, form the process group (gloo where there is no GPU), all-reduce a "
                         "one per rank, print {n_gpus, rccl_ranks} on rank 0 and stop: proves `--gpus N` reaches N ranks")
    ap.add_argument("--steps", type=int, default=120,
                    help="frames in the timed interval (the pipeline is primed before it and kept full behind it)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=8192, help="frame edge in pixels (per GPU)")
    ap.add_argument("--depth", type=int, default=16, choices=(8, 16))
    ap.add_argument("--kind", default="photo")
    ap.add_argument("--streams", type=int, default=16, help="contexts (one HIP stream each), frames-per-launch-group frames in flight on each")
    ap.add_argument("--rans-waves", type=int, default=5, choices=(4, 5, 6),
                    help="entropy-stage form of the frame loop and the shard leg, see hydamd_set_rans_waves: 5 = one lane per group, a "
                         "wavefront per LF group (tables in LDS sized by the clustering scheme since round 5); 4 = one wave per group "
                         "(lowest single-frame latency); 6 = rounds 3-4's packed-table variant of 5, now another name of 5")
    ap.add_argument("--no-content", action="store_true",
                    help="frame mode: leave out the rows for SURVEY 8(d)'s other two synthetic inputs (smooth, noise)")
    ap.add_argument("--lf-coder", default="on", choices=("on", "off"),
                    help="code the LF coefficient streams on the GPU inside the timed loop (default) or leave them out")
    ap.add_argument("--collective", default="gather", choices=("gather", "all-gather"),
                    help="N > 1: bring each frame's sections to rank 0 only (default) or to every rank")
    ap.add_argument("--gather-every", type=int, default=1, help="N > 1: frames whose blobs travel in one RCCL gather")
    ap.add_argument("--exchange", action="store_true",
                    help="frame mode: also export every frame's results as a blob and gather the blobs with RCCL (one frame per launch "
                         "group then); without it N ranks code their own frames and only the clock is shared")
    ap.add_argument("--frames-per-launch", type=int, default=2,
                    help="frame mode: independent frames coded as ONE launch group per context (hydamd_encode_image_batch): the serial "
                         "rANS chains of the group run side by side, so a stream is held for one chain's 2.5 ms per group instead of per frame")
    ap.add_argument("--mode", default="frame", choices=("frame", "shard", "batch"),
                    help="frame: one 8192x8192 RGB16 frame per GPU per step (BASELINE configs[2], the contract line); "
                         "shard: ONE 16384x16384 RGB8 frame per step, its LF groups spread over the GPUs (configs[3]); "
                         "batch: 64 independent 3840x2160 RGB8 frames through the drop-in API, frame i on GPU i mod N (configs[4])")
    ap.add_argument("--assemble", default="device", choices=("device", "device-pinned", "host"),
                    help="--mode shard: where the frame is put together — on the assembling rank's GPU, into device memory and then "
                         "one D2H copy (default), or written straight into pinned host memory by the kernel; or on its host from "
                         "the blobs (round 2)")
    ap.add_argument("--shard-depth", type=int, default=0,
                    help="--mode shard: frames in flight (a context per frame and rank, sized for the rank's LF groups); 0 = 8 on one GPU "
                         "(scan 4 / 6 / 8 / 10 / 12: 2.45 / 1.97 / 1.90 / 1.95 / 1.86 ms per 16K frame), 12 on two, 16 on four and more: a "
                         "rank's transform stage shrinks with N, its 2.5 ms rANS chains do not")
    ap.add_argument("--frames", type=int, default=64, help="--mode batch: frames in the batch")
    ap.add_argument("--threads", type=int, default=10, help="--mode batch: host threads (encoders) per GPU (scan in profiles/r04_batch_host_scan.txt: 8 / 10 / 12 / 16 threads 1456 / 1568 / 1521 / 1376 frames/s)")
    ap.add_argument("--no-legs", action="store_true",
                    help="frame mode: leave out the configs[3] / configs[4] legs (shard_16k, batch_4k) and the LF-off leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-api", action="store_true")
    ap.add_argument("--no-bind", action="store_true",
                    help="leave the process where the scheduler put it instead of binding it to the CPUs of the GPU's NUMA node")
    return ap.parse_args()


def launch_ranks(args, argv=None):
    """`--gpus N` names the job's size.  Inside a process group (torch.distributed.run exported WORLD_SIZE) it must be the
    group's; outside one, N > 1 means: start the N ranks here — the same command under torch.distributed.run, one rank per
    GPU of this node, rendezvous on 127.0.0.1 — and return their exit status.  Returns None when this process is a rank
    (or the whole job, N = 1) and should carry on."""
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is not None:
        world = int(env_world)
        if args.gpus is None:
            args.gpus = world
        if args.gpus != world:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but the process group has {world} ranks "
                             f"(WORLD_SIZE): launch with --nproc-per-node {args.gpus} or leave --gpus to the launcher")
        return None
    if args.gpus is None:
        args.gpus = 1
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be at least 1")
    if args.gpus == 1:
        return None
    import socket
    import subprocess

    with socket.socket() as sock:  # a free rendezvous port (two benches on one node must not meet)
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), HYDAMD_BENCH_SELF_LAUNCHED="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(sys.argv[1:] if argv is None else argv)
    print(f"bench.py: --gpus {args.gpus} outside a process group: starting {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def check_world(args, dist, device=None):
    """after init_process_group: the group IS the job --gpus named, and every rank answers (-> rccl_ranks)"""
    import torch

    world = dist.get_world_size()
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the process group has {world} ranks")
    one = torch.ones(1, dtype=torch.int32, device=device if device is not None else "cpu")
    dist.all_reduce(one)
    ranks = int(one.item())
    if ranks != world:
        raise SystemExit(f"bench.py: an all-reduce of ones over {world} ranks returned {ranks}")
    return ranks


# what a leg's result keeps in the frame-mode line
LEG_KEYS = ("value", "unit", "ms_per_step", "frames_per_s", "n_gpus", "rccl_ranks", "steps", "scaling", "config", "frame_bytes", "frame_md5",
            "frames_checked", "assembled_frames_identical_to_host_assembly", "host_ms_per_frame", "frac_of_hbm_read_roofline",
            "frame0_identical_to_reference", "threads_agree_with_single_thread_run", "frames_per_s_each_round", "leg_wall_s", "error",
            "sections_identical_to_single_context_run", "frames_per_s_one_frame_per_launch_group")
_RANK_ENV = ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "GROUP_WORLD_SIZE", "ROLE_RANK", "ROLE_WORLD_SIZE",
             "ROLE_NAME", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS",
             "TORCHELASTIC_USE_AGENT_STORE", "TORCHELASTIC_ERROR_FILE", "TORCH_NCCL_ASYNC_ERROR_HANDLING", "HYDAMD_BENCH_SELF_LAUNCHED",
             "HYDAMD_DEVICE", "HYDAMD_DEVICES", "HYDAMD_BENCH_FORCE_PG")


def child_job(gpus, argv, timeout=300):
    """A leg of an N-rank job as a JOB OF ITS OWN: `bench.py --gpus N <argv>` started from rank 0 once the main process group
    is gone, outside any rank environment (it launches its N ranks itself), under a time limit.  A leg that needs every rank —
    the sharded 16K frame, the 4K batch — must not run inside the job whose headline it accompanies: a rank that fails there
    leaves the others in a collective, the group's watchdog ends the job, and the line is lost.  -> the child's result line as
    a dict, or {"error": ...}."""
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in _RANK_ENV}
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(gpus)] + list(argv)
    import signal

    # a session of its own: on overrun the launcher, torch.distributed.run and every rank go together
    pr = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, start_new_session=True)
    try:
        stdout, stderr = pr.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(pr.pid, signal.SIGKILL)
        except OSError:
            pass
        pr.communicate()
        return {"error": f"no result within {timeout} s: {' '.join(cmd)}"}
    line = next((l for l in reversed(stdout.splitlines()) if l.startswith("{")), None)
    if line is None:
        return {"error": f"exit status {pr.returncode}: {(stderr or stdout)[-400:]}"}
    try:
        return json.loads(line)
    except ValueError as exc:
        return {"error": f"unparsable result line ({exc}): {line[:200]}"}


def dry_run_launch(args):
    import torch
    import torch.distributed as dist

    for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29535"), ("RANK", "0"), ("WORLD_SIZE", "1")):
        os.environ.setdefault(k, v)
    local = int(os.environ.get("LOCAL_RANK", "0"))
    gpu = torch.cuda.is_available()
    if gpu:
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group("gloo")
    ranks = check_world(args, dist, torch.device("cuda", local) if gpu else None)
    out = {"dry_run_launch": True, "n_gpus": dist.get_world_size(), "rccl_ranks": ranks, "backend": "nccl" if gpu else "gloo",
           "launched_by": "bench.py itself" if os.environ.get("HYDAMD_BENCH_SELF_LAUNCHED") else
                          "the caller's torch.distributed.run" if os.environ.get("TORCHELASTIC_RUN_ID") else "nobody: a single process"}
    rank = dist.get_rank()
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        emit(out)


def cpu_baseline(host_img):
    """hydrium's own CPU path (oracle/_ref build, -Os as shipped) on this box's host cores, or the
    CPU port when the prebuilt reference did not travel.  Reported next to the GPU number."""
    from hydrium_amd import api
    from oracle import refprobe

    h, w, _ = host_img.shape
    if refprobe.available():
        lib = refprobe.reference_library()
        reps = 4
        t0 = time.perf_counter()
        for _ in range(reps):
            data = api.encode_image(lib, host_img)
        dt = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        api.encode_image(refprobe.reference_library(optimised=True), host_img)
        dt_o2 = time.perf_counter() - t0
        return {"value": round(w * h / dt / 1e6, 3), "unit": "Mpixel/s", "cores": 1, "kind": "reference",
                "sample": f"the whole {w}x{h} workload frame, {reps} times through hyd_send_tile in one-frame mode "
                          f"(reference sources built gcc -Os as shipped, {dt * reps:.1f} s of CPU work)",
                "value_O2_build": round(w * h / dt_o2 / 1e6, 3),
                "bytes": len(data), "md5": hashlib.md5(data).hexdigest()}
    from oracle import binding as orc

    crop = host_img[:4096, :4096].copy()
    t0 = time.perf_counter()
    nbytes, _ = orc.hot_path_image(crop)
    dt = time.perf_counter() - t0
    return {"value": round(crop.shape[0] * crop.shape[1] / dt / 1e6, 3), "unit": "Mpixel/s", "cores": 1, "kind": "port",
            "sample": f"hot path only (oracle/hyd_oracle.c) on the top-left 4096x4096 of the workload frame, {dt:.1f} s",
            "bytes": int(nbytes)}


def shard_leg(args, steps, warmup, size=16384, kind=None, assemble="device"):
    """BASELINE configs[3]: ONE size x size RGB8 frame per step, its LF groups sharded over the ranks; presets
    numbered across the frame, alphabet floor exchanged on the device, one gather of blobs to the assembling
    rank (step mod N), where the frame is put together ON THE GPU (hydamd_assembler_*) and lands in pinned
    host memory.  torch.distributed must be initialised.  Returns the result dict on rank 0, None elsewhere."""
    import numpy as np
    import torch
    import torch.distributed as dist

    from hydrium_amd import api, device, multigpu, sharding, synth

    rank, world = dist.get_rank(), dist.get_world_size()
    local = int(os.environ.get("LOCAL_RANK", "0"))
    kind = kind or args.kind
    W = H = size
    depth = 8
    lfx, lfy = -(-W // 2048), -(-H // 2048)
    n_lf = lfx * lfy
    parts = sharding.partition_lf_groups(n_lf, world)
    mine = parts[rank]
    # this rank's LF groups are whole rows of LF groups when N divides the row count: its slab is a band
    rows = sorted({lf // lfx for lf in mine})
    y0 = rows[0] * 2048 if rows else 0
    band_h = (min(H, (rows[-1] + 1) * 2048) - y0) if rows else 8
    dev = torch.device("cuda", local)
    slab = synth.make_image(kind, W, band_h, depth, x0=0, y0=y0, device=dev)
    origin = lambda lf: ((lf // lfx) * 2048 - y0, (lf % lfx) * 2048)
    depth_in_flight = max(2, args.shard_depth) if args.shard_depth else (8 if world == 1 else 12 if world == 2 else 16)
    shards = [multigpu.Shard(local, mine, W, H) for _ in range(depth_in_flight)]
    for sh in shards:
        if sh.ctx:
            sh.ctx.set_rans_waves(args.rans_waves)
            # the LF coder in the context's own stream (its code construction rides in the chain kernel's launch): one
            # stream per frame in flight, as in frame mode — side streams alias onto the hardware queues of other frames
            sh.ctx.set_lf_coder(2)
    engines = [multigpu.GpuShardEngine(sh, slab, origin) for sh in shards]
    on_device = assemble != "host"
    assemblies = [multigpu.FrameAssembly(local, W, H, None, pinned=(assemble == "device-pinned")) if on_device else None
                  for _ in range(depth_in_flight)]
    md = api.HYDImageMetadata(W, H, 0, -1, -1)

    # first frame: exact sizes, the host assembler's bytes for the MD5 (tests tie those to the reference), the blob capacity
    with torch.cuda.stream(engines[0].stream):
        blobs = multigpu.choreograph_frame(engines[0], parts)
    first = device.frame_from_blobs(md, blobs) if rank == 0 else None
    biggest = torch.tensor([max((len(b) for b in blobs), default=0) if blobs else 0], dtype=torch.int64, device=dev)
    dist.broadcast(biggest, src=0)
    cap = int(int(biggest.item()) * 1.25) + 65536
    pinned = [torch.empty(cap, dtype=torch.uint8).pin_memory() for _ in range(world)] if not on_device else None

    asm_ms, d2h_ms = [], []

    def assemble_on_host(rows_dev, want_md5):
        """the round-2 path: blobs -> pinned host memory -> hydamd_frame_from_blobs"""
        t_a = time.perf_counter()
        heads = torch.stack([r[:64] for r in rows_dev]).cpu().numpy()
        sizes = [int(device.blob_header(heads[k].tobytes())["total_bytes"]) for k in range(len(rows_dev))]
        for r, p, n in zip(rows_dev, pinned, sizes):
            p[:n].copy_(r[:n], non_blocking=True)
        torch.cuda.current_stream().synchronize()
        t_b = time.perf_counter()
        view, release = device.frame_from_blobs(md, [p[:n].numpy() for p, n in zip(pinned, sizes)], raw=True)
        t_c = time.perf_counter()
        digest = hashlib.md5(view).hexdigest() if want_md5 else None
        release()
        d2h_ms.append((t_b - t_a) * 1e3)
        asm_ms.append((t_c - t_b) * 1e3)
        return digest

    handles = []
    landing = []  # assemblies whose frame is on its way to the host
    retries = 0
    md5s = {}
    checking = [True]  # frames assembled during warm-up are hashed and compared with the first frame; timed ones are not
    host_ms = [0.0, 0.0]

    def enqueue(i):
        t_a = time.perf_counter()
        e = engines[i % depth_in_flight]
        with torch.cuda.stream(e.stream):
            handles.append((i, multigpu.enqueue_frame(e, parts, cap, None, i % world, assemblies[i % depth_in_flight])))
        host_ms[0] += time.perf_counter() - t_a

    def collect():
        nonlocal retries
        i, h = handles.pop(0)
        again, rows_dev = multigpu.collect_frame(h)
        t_a = time.perf_counter()
        retries += int(again)
        digest = None
        if rows_dev is not None and not again:
            with torch.cuda.stream(h["engine"].stream):
                if on_device:
                    # the finished codestream -> pinned host memory: the copy runs beside the next frame's kernels and is
                    # waited for when that frame is collected (or when the clock stops)
                    h["assembly"].start_copy()
                    landing.append(h["assembly"])
                    if checking[0]:
                        digest = hashlib.md5(h["assembly"].to_host().cpu().numpy()).hexdigest()
                else:
                    digest = assemble_on_host(rows_dev, checking[0])
        while len(landing) >= depth_in_flight:  # an assembly's buffers are reused depth_in_flight frames later: its copy must have landed by then
            landing.pop(0).to_host()
        host_ms[1] += time.perf_counter() - t_a
        return i, digest

    # warm-up, then a steady-state interval: depth-1 frames are in flight when the clock starts and when it stops
    seq = 0
    for _ in range(warmup + depth_in_flight - 1):
        enqueue(seq)
        seq += 1
        if len(handles) == depth_in_flight:
            i, digest = collect()
            if digest is not None:
                md5s[i] = digest
    checking[0] = False
    host_ms[0] = host_ms[1] = 0.0
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        enqueue(seq)
        seq += 1
        collect()
    while landing:
        landing.pop(0).to_host()
    torch.cuda.synchronize()
    dist.barrier()
    dt = time.perf_counter() - t0
    host_timed = (host_ms[0], host_ms[1])
    if os.environ.get("HYDAMD_DEBUG_D2H"):  # per frame: host time issuing the copy, the copy's own duration, host time waiting for it
        dbg = multigpu.FrameAssembly._dbg
        issue = [x for x in dbg if len(x) == 4][-16:]
        print("D2H issue, host ms:", " ".join(f"{x[0] * 1e3:.2f}" for x in issue), file=sys.stderr)
        print("D2H copy, stream ms:", " ".join(f"{x[1].elapsed_time(x[2]):.2f}" for x in issue), file=sys.stderr)
        print("D2H wait, host ms:", " ".join([f"{x[0] * 1e3:.2f}" for x in dbg if len(x) == 1][-16:]), file=sys.stderr)
    checking[0] = True
    while handles:
        i, digest = collect()
        if digest is not None:
            md5s[i] = digest
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    # every rank's assembly times and its MD5s, for the line
    gathered = [None] * world
    dist.all_gather_object(gathered, dict(asm=asm_ms[-8:], d2h=d2h_ms[-8:], md5=list(md5s.values()), retries=retries,
                                          enq=host_timed[0] / max(steps, 1) * 1e3, col=host_timed[1] / max(steps, 1) * 1e3))
    out = None
    if rank == 0:
        want = hashlib.md5(first).hexdigest()
        seen = [m for g in gathered for m in g["md5"]]
        asm = [a for g in gathered for a in g["asm"]]
        out = {
            "metric": "Mpixel/s encode (16K RGB8 frame sharded by LF group)", "mode": "shard",
            "value": round(W * H * steps / dt / 1e6, 1), "unit": "Mpixel/s", "n_gpus": world,
            "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"one {W}x{H} RGB{depth} '{kind}' frame per step (BASELINE configs[3]), {n_lf} LF groups "
                                   f"in raster blocks of {len(parts[0])} per GPU, presets numbered across the frame; the finished "
                                   "codestream is in pinned host memory when a step ends",
                       "exchange": "all-gather of int32 alphabet maxima on the device; one RCCL gather of the shards' blobs "
                                   "(hydamd_export_frame) to the assembling rank, which rotates with the step",
                       "assembly": {"device-pinned": "on the assembling rank's GPU (hydamd_assembler_*), written straight into pinned host memory",
                                    "device": "on the assembling rank's GPU (hydamd_assembler_*), then one D2H copy",
                                    "host": "blobs copied to the host, hydamd_frame_from_blobs (round 2's path)"}[assemble],
                       "frames_in_flight": depth_in_flight, "blob_capacity_bytes": cap, "parallelism": f"{world}-way LF-group shard"},
            "frame_bytes": len(first), "frame_md5": want,
            "frames_checked": len(seen),
            "assembled_frames_identical_to_host_assembly": bool(seen) and all(m == want for m in seen),
            "host_ms_per_frame": {"enqueue": round(max(g["enq"] for g in gathered), 3),
                                  "after_sync": round(max(g["col"] for g in gathered), 3)},
            "reruns_after_buffer_overflow": sum(g["retries"] for g in gathered),
            "frac_of_hbm_read_roofline": round((W * H * steps / dt) / (HBM_PEAK_GBS * 1e9 * world / 3), 5),
        }
        if not on_device:
            out["host_assembly_ms"] = round(sum(asm) / max(len(asm), 1), 3)
            out["blobs_to_host_ms"] = round(sum(a for g in gathered for a in g["d2h"]) / max(sum(len(g["d2h"]) for g in gathered), 1), 3)
    dist.barrier()
    torch.cuda.synchronize()
    # torch's caching allocator keeps the blocks it handed out under the contexts' streams in per-stream
    # pools: they have to go back to the driver before those streams are destroyed with their contexts
    handles.clear()
    for a in assemblies:
        if a is not None:
            a.close()
    del blobs, pinned, engines, assemblies
    for sh in shards:
        if sh.ctx:
            sh.ctx.__dict__.pop("_floor_keep", None)
    import gc

    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    for sh in shards:
        sh.close()
    return out


def run_shard(args):
    import torch
    import torch.distributed as dist

    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if not args.no_bind:
        from hydrium_amd import placement

        placement.bind_near_gpu(local)
    for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29534"), ("RANK", "0"), ("WORLD_SIZE", "1")):
        os.environ.setdefault(k, v)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    ranks = check_world(args, dist, torch.device("cuda", local))
    out = shard_leg(args, args.steps, args.warmup, args.size if args.size != 8192 else 16384, assemble=args.assemble)
    if out is not None:
        out["rccl_ranks"] = ranks
    dist.barrier()
    torch.cuda.synchronize()
    dist.destroy_process_group()  # RCCL has seen the contexts' streams: it goes first
    if out is not None:
        emit(out)


def batch_device_leg(args, frames, contexts=16, rounds=4, per_group=8):
    """BASELINE configs[4] WITHOUT PCIe: the batch of independent 3840x2160 RGB8 frames already resident in HBM
    ("synthetic RGB tiles"), frame i on GPU i mod N, each rank's frames dealt to `contexts` device contexts `per_group`
    at a time (hydamd_encode_image_batch: four 4K frames = sixteen LF groups = one launch group, the shape of an 8K
    frame; per_group 1 = hydamd_encode_image, one call and one launch group per frame), every frame's finished sections
    left in HBM.  After the clock stops the sections every context holds are compared, byte for byte, with those of the
    same pictures coded one frame at a time on one context."""
    import torch
    import torch.distributed as dist

    from hydrium_amd import device, synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 and dist.is_initialized()
    w, h = 3840, 2160
    distinct = 8
    imgs = [synth.make_image("photo", w, h, 8, seed=1234 + k, device=torch.device("cuda", local)) for k in range(distinct)]
    mine = list(range(rank, frames, world))
    S = max(1, min(contexts, len(mine) or 1))
    ctxs = [device.DeviceContext(local, 4 * per_group, 0) for _ in range(S)]
    for c in ctxs:
        c.set_rans_waves(5)
        c.set_lf_coder(2)
    want = {}
    for k in range(distinct):  # reference sections: one context, one frame at a time
        ctxs[0].encode_image_tensor(imgs[k])
        ctxs[0].sync()
        want[k] = bytes(ctxs[0].read_payload())

    def run(G):
        # several frames per launch group: the loop's regime, form 6 (+1.5 %); a frame per group: four chains per launch,
        # their own duration counts, form 5 (+6 %)
        for c in ctxs:
            c.set_rans_waves(args.rans_waves if G > 1 and args.rans_waves >= 5 else 5)
        groups = [mine[i:i + G] for i in range(0, len(mine), G)]
        for c in ctxs:  # every context's buffers touched once
            c.encode_image_batch([imgs[0]] * G) if G > 1 else c.encode_image_tensor(imgs[0])
        for c in ctxs:
            c.sync()
        dts, last = [], {}
        for _ in range(rounds):
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i, grp in enumerate(groups):
                if G > 1 and len(grp) == G:
                    ctxs[i % S].encode_image_batch([imgs[f % distinct] for f in grp])
                    last[i % S] = grp
                else:
                    for f in grp:
                        ctxs[i % S].encode_image_tensor(imgs[f % distinct])
                        last[i % S] = [f]
            for c in ctxs:
                c.sync()
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier()
            dts.append(time.perf_counter() - t0)
        ok = all(bytes(ctxs[k].read_payload()) == b"".join(want[f % distinct] for f in grp) for k, grp in last.items())
        if use_dist:
            tt = torch.tensor(dts, dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dts = [float(x) for x in tt.tolist()]
        timed = sorted(dts[1:])
        return timed[len(timed) // 2], dts, ok

    dt, dts, ok = run(per_group)
    dt1, dts1, ok1 = run(1) if per_group > 1 else (dt, dts, ok)
    for c in ctxs:
        c.close()
    if rank != 0:
        return None
    return {"value": round(frames * w * h / dt / 1e6, 1), "unit": "Mpixel/s", "frames_per_s": round(frames / dt, 1),
            "ms_per_step": round(dt / frames * 1e3, 4), "n_gpus": world, "steps": frames, "scaling": "strong",
            "frames_per_s_each_round": [round(frames / x, 1) for x in dts[1:]],
            "frames_per_s_one_frame_per_launch_group": round(frames / dt1, 1),
            "frac_of_hbm_read_roofline": round(frames * w * h * 3 / dt / (HBM_PEAK_GBS * 1e9), 5),
            "config": {"workload": f"{frames} independent {w}x{h} RGB8 'photo' frames (BASELINE configs[4]) resident in HBM, "
                                   f"dealt to {S} device contexts per GPU {per_group} at a time (one hydamd_encode_image_batch call = one "
                                   f"launch group of {4 * per_group} LF groups), frame i on GPU i mod {world}; sections + coded LF streams "
                                   "of every frame left in HBM (no PCIe in the timed region: the host-pointer form is batch_4k); wall clock "
                                   "around the whole batch, fill and drain included",
                       "contexts_per_gpu": S, "frames_per_launch_group": per_group},
            "sections_identical_to_single_context_run": bool(ok and ok1)}


def batch_leg(args, frames, threads, rounds=4):
    """BASELINE configs[4]: a batch of independent 3840x2160 RGB8 frames through the drop-in API
    (hyd_encoder_new .. hyd_send_tile .. hyd_flush from host memory), frame i on GPU i mod N, several
    encoder threads per GPU, no collective in the data path.  HYDAMD_DEVICE must name this rank's GPU
    before the library is first used.  Returns the result dict on rank 0, None elsewhere."""
    import ctypes
    import threading

    import numpy as np
    import torch
    import torch.distributed as dist

    from hydrium_amd import api, synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    use_dist = world > 1 and dist.is_initialized()
    w, h = 3840, 2160
    distinct = 8  # frame i shows picture i mod 8 (seed 1234 + i mod 8)
    imgs = [np.ascontiguousarray(synth.make_image("photo", w, h, 8, seed=1234 + k, device="cuda").cpu().numpy()) for k in range(distinct)]
    lib = api.Library()
    mine = list(range(rank, frames, world))
    md5 = {}
    for k in range(min(distinct, 2)):  # warm the library, the context pool and the reference MD5s
        md5[k] = hashlib.md5(api.encode_image(lib, imgs[k])).hexdigest()
    T = max(1, min(threads, len(mine) or 1))
    got = {}
    # every thread keeps one output buffer, as a caller encoding many frames would; the codestreams are kept and
    # hashed after the clock has stopped (verification is not part of the encode)
    bufs = [(ctypes.c_uint8 * (16 << 20))() for _ in range(T)]

    def work(t):
        for f in mine[t::T]:
            got[f] = api.encode_image(lib, imgs[f % distinct], out_buf=bufs[t])

    dts = []
    for _ in range(rounds):  # the first round is the warm-up (contexts created and parked); the median of the others counts
        ts = [threading.Thread(target=work, args=(t,)) for t in range(T)]
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        dts.append(time.perf_counter() - t0)
    if use_dist:
        tt = torch.tensor(dts, dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dts = [float(x) for x in tt.tolist()]
    timed = sorted(dts[1:]) if len(dts) > 1 else dts
    dt = timed[len(timed) // 2]
    consistent = all(hashlib.md5(got[f]).hexdigest() == md5[f % distinct] for f in got if f % distinct in md5)
    if rank != 0:
        return None
    want = None
    with open(os.path.join(ROOT, "tests", "golden", "manifest.json")) as f:
        for e in json.load(f)["files"]:
            if (e["kind"], e["width"], e["height"], e["depth"], e["shift"]) == ("photo", w, h, 8, -1):
                want = e["md5"]
    return {
        "metric": "Mpixel/s encode (batch of 3840x2160 RGB8 frames, drop-in API)", "mode": "batch",
        "value": round(frames * w * h / dt / 1e6, 1), "unit": "Mpixel/s", "frames_per_s": round(frames / dt, 1),
        "n_gpus": world, "steps": frames, "warmup": 1, "ms_per_step": round(dt / frames * 1e3, 4),
        "frames_per_s_each_round": [round(frames / x, 1) for x in dts[1:]],
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{frames} independent {w}x{h} RGB8 'photo' frames (BASELINE configs[4]), host pixels "
                               "through hyd_send_tile in one-frame mode, PCIe, read-back and frame assembly inclusive; "
                               "codestreams kept and compared with the single-thread run after the clock stops",
                   "threads_per_gpu": T, "parallelism": f"frame i on GPU i mod {world}, no collective"},
        "frame0_md5": md5.get(0), "frame0_md5_in_golden_manifest": want,
        "frame0_identical_to_reference": md5.get(0) == want if want else None,
        "threads_agree_with_single_thread_run": consistent,
    }


def run_batch(args):
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ["HYDAMD_DEVICE"] = str(local)
    os.environ.setdefault("HYDAMD_CONTEXT_CACHE", str(max(4, args.threads)))
    torch.cuda.set_device(local)
    if not args.no_bind:
        from hydrium_amd import placement

        placement.bind_near_gpu(local)
    ranks = 1
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        ranks = check_world(args, dist, torch.device("cuda", local))
    out = batch_leg(args, args.frames, args.threads)
    if out is not None:
        out["rccl_ranks"] = ranks
    if world > 1:
        dist.destroy_process_group()
    if out is not None:
        emit(out)


def loop_rate(ctxs, groups, ext, W, H, FPL, prime_groups, timed_groups):
    """the frame loop of `main`, stripped: contexts take launch groups in turn; HIP events at the end of every group's stream;
    -> sustained Mpixel/s over `timed_groups` groups per stream behind `prime_groups` of priming"""
    import torch

    S = len(ctxs)
    evs = []
    n = (prime_groups + timed_groups + 1) * S
    for i in range(n):
        k = i % S
        ctxs[k].encode_image_batch(groups[k]) if FPL > 1 else ctxs[k].encode_image_tensor(groups[k][0])
        e = torch.cuda.Event(enable_timing=True)
        e.record(ext[k])
        evs.append(e)
    for c in ctxs:
        c.sync()
    torch.cuda.synchronize()
    done = [evs[0].elapsed_time(e) for e in evs]
    a, b = prime_groups * S, (prime_groups + timed_groups) * S
    t0, t1 = sum(done[a - S:a]) / S, sum(done[b - S:b]) / S
    return W * H * FPL * timed_groups * S / ((t1 - t0) * 1e-3) / 1e6


def content_row(args, kind, local):
    """SURVEY 8(d) / BASELINE.md 3 name three synthetic inputs: 'photo' is the contract line; this is the same set of figures
    for one of the other two (8192x8192 RGB16): the pipelined loop's sustained rate, one frame alone, the drop-in API end to
    end, each beside the CPU reference on one core over the same frame, with the byte-equality verdict."""
    import ctypes

    import numpy as np
    import torch

    from hydrium_amd import api, device, synth
    from oracle import refprobe

    W = H = args.size
    S, FPL = max(1, args.streams), max(1, args.frames_per_launch)
    lfg = (-(-W // 2048)) * (-(-H // 2048))
    dev = torch.device("cuda", local)
    img = synth.make_image(kind, W, H, args.depth, device=dev)
    torch.cuda.synchronize()
    row = {"kind": kind}
    ctxs = [device.DeviceContext(local, lfg * FPL, 0) for _ in range(S)]
    try:
        ext = [torch.cuda.ExternalStream(c.get_stream()) for c in ctxs]
        groups = [[img] * FPL for _ in ctxs]
        for k, c in enumerate(ctxs):
            c.set_rans_waves(args.rans_waves)
            c.set_lf_coder(2)
            c.encode_image_batch([img] * FPL) if FPL > 1 else c.encode_image_tensor(img)
            if k == 0:
                # the first context finds out what kind of content this queue holds: a frame that outgrows the default
                # buffers (noise: 2.9 symbols per pixel) is rerun inside this wait; the others — idle so far — take the
                # hint at their first frame (hydamd_begin_frame, round 6) instead of each finding out by itself
                c.sync()
        for c in ctxs:
            c.sync()
        row["overflow_reruns_first_frame"] = int(sum(c.overflow_reruns() for c in ctxs))
        row["contexts_enlarged_ahead_of_their_first_frame"] = int(sum(c.grown_ahead() for c in ctxs))
        rate = loop_rate(ctxs, groups, ext, W, H, FPL, 4, 6)
        row["Mpixel/s"] = round(rate, 1)
        row["ms_per_frame_in_the_loop"] = round(W * H / rate / 1e3, 4)
        row["symbols_per_pixel"] = round(sum(int(ctxs[0].read_symbol_counts(s).sum()) for s in range(lfg)) / (W * H), 4)
        row["section_bytes"] = ctxs[0].payload_size() // FPL
        c0 = ctxs[0]
        for form, key in ((max(5, args.rans_waves), "lane_form"), (4, "wave_form")):
            c0.set_rans_waves(form)
            c0.set_lf_coder(2 if form >= 5 else 1)
            c0.encode_image_tensor(img)
            c0.sync()
            c0.profile(True)
            t = time.perf_counter()
            for _ in range(3):
                c0.encode_image_tensor(img)
                c0.sync()
            t = (time.perf_counter() - t) / 3
            k = {n_: round(ms / max(n, 1), 4) for n_, (ms, n) in c0.profile_read().items()}
            c0.profile(False)
            row["single_frame_" + key] = {"ms_per_frame": round(t * 1e3, 4), "transform_ms": k.get("transform_tokenize"), "chains_ms": k.get("rans_encode")}
    finally:
        for c in ctxs:
            c.close()
    arr = img.cpu().numpy()
    host = np.ascontiguousarray(arr.view(np.uint16) if args.depth == 16 else arr)
    del img
    torch.cuda.empty_cache()
    lib = api.Library()
    big = (ctypes.c_uint8 * (256 << 20))()  # noise compresses to ~2 bytes per pixel
    api.encode_image(lib, host, out_buf=big)
    times = []
    for _ in range(3):
        t = time.perf_counter()
        data = api.encode_image(lib, host, out_buf=big, in_place=True)
        times.append(time.perf_counter() - t)
    data = bytes(data)
    row["api_end_to_end"] = {"ms": round(sorted(times)[1] * 1e3, 1), "bytes": len(data), "md5": hashlib.md5(data).hexdigest()}
    if refprobe.available() and not args.no_cpu_baseline:
        t = time.perf_counter()
        ref = api.encode_image(refprobe.reference_library(), host, out_buf_size=64 << 20)
        t = time.perf_counter() - t
        row["cpu_baseline"] = {"value": round(W * H / t / 1e6, 3), "unit": "Mpixel/s", "cores": 1, "kind": "reference",
                               "sample": f"the whole frame once through hyd_send_tile ({t:.1f} s of CPU work)", "bytes": len(ref)}
        row["api_end_to_end"]["identical_to_cpu_reference"] = ref == data
    trim = getattr(lib.dll, "hydamd_trim_cache", None)
    if trim is not None:
        trim.restype = None
        trim()
    return row


_MULTI_DEVICE_CLIENT = r"""
import ctypes, hashlib, json, sys, time
import numpy as np
import torch
from hydrium_amd import api, synth
size, verify = int(sys.argv[1]), int(sys.argv[2])
t = synth.make_image("photo", size, size, 8, device="cuda")
torch.cuda.synchronize()
img = np.ascontiguousarray(t.cpu().numpy())
del t
torch.cuda.empty_cache()
lib = api.Library()
buf = (ctypes.c_uint8 * (96 << 20))()
t0 = time.perf_counter()
api.encode_image(lib, img, out_buf=buf)
first = time.perf_counter() - t0
times = []
for _ in range(5):
    t0 = time.perf_counter()
    data = api.encode_image(lib, img, out_buf=buf, in_place=True)
    times.append(time.perf_counter() - t0)
data = bytes(data)
print("RESULT " + json.dumps({"ms": round(sorted(times)[2] * 1e3, 2), "ms_each": [round(x * 1e3, 2) for x in times],
                              "first_call_ms": round(first * 1e3, 1), "bytes": len(data), "md5": hashlib.md5(data).hexdigest()}))
"""


def api_multi_device_leg(ndev, size=16384):
    """The multi-device tile scheduler INSIDE the C library (csrc/host/encoder.c finish_frame_multi): one process, one
    encoder, `hyd_send_tile` from host memory, the frame's LF groups dealt to every device of HYDAMD_DEVICES, the frame
    assembled on the first one from peer reads.  On a box with ONE GPU the list names it four times (four contexts, every
    cross-context step taken, only the xGMI hop missing).  A process of its own: the device list is read once per process,
    and this one's API legs are pinned to its own GPU.  A second run with HYDAMD_VERIFY_PEERS=1 checks every shard's view
    on both sides of the peer read."""
    import subprocess

    devices = ",".join(str(d) for d in range(ndev)) if ndev > 1 else "0,0,0,0"
    out = {"devices": devices, "aliased": ndev <= 1,
           "workload": f"one {size}x{size} RGB8 'photo' frame (BASELINE configs[3]) from host memory through hyd_send_tile, one-frame mode; "
                       "PCIe, peer reads, assembly on the first device and read-back inclusive; median of 5 frames after the first"}
    for verify in (0, 1):
        env = dict(os.environ, PYTHONPATH=ROOT, HYDAMD_DEVICES=devices, HYDAMD_VERIFY_PEERS=str(verify))
        for k in ("HYDAMD_DEVICE", "RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
            env.pop(k, None)
        r = subprocess.run([sys.executable, "-c", _MULTI_DEVICE_CLIENT, str(size), str(verify)], capture_output=True, text=True, env=env, timeout=300)
        line = next((l for l in r.stdout.splitlines() if l.startswith("RESULT ")), None)
        if r.returncode or not line:
            out["error" if not verify else "verify_error"] = (r.stderr or r.stdout)[-400:]
            break
        res = json.loads(line[7:])
        if not verify:
            out.update(res)
            out["Mpixel/s"] = round(size * size / res["ms"] / 1e3, 1)
        else:
            out["with_peer_views_verified"] = {"ms": res["ms"], "same_file": res["md5"] == out.get("md5")}
    return out


_INPROCESS_CLIENT = r"""
import hashlib, json, sys, time
import numpy as np
import torch
from hydrium_amd import device, synth
ndev, size, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
devices = list(range(ndev)) if ndev > 1 else [0, 0, 0, 0]
n = len(devices)
D = 4 if ndev <= 1 else 8                       # frames in flight: one HydAmdMulti each (n contexts, n streams)
pics = {d: synth.make_image("photo", size, size, 8, device=torch.device("cuda", d)) for d in sorted(set(devices))}
for d in pics:
    torch.cuda.synchronize(d)
origins = [pics[d] for d in devices]
frames = [device.MultiFrame(devices, size, size) for _ in range(D)]
for i, m in enumerate(frames):                  # first use: contexts' lazy allocations, the peer pairs' first-use verification
    m.encode(origins, assembling_shard=i % n)
sizes = {m.result() for m in frames}
def run(steps, to_host):
    pinned = torch.empty(max(sizes) + 4096, dtype=torch.uint8).pin_memory().numpy() if to_host else None
    t0 = time.perf_counter()
    for i in range(steps + D):
        m = frames[i % D]
        if i >= D:
            m.read(pinned) if to_host else m.result()
        if i < steps:
            m.encode(origins, assembling_shard=i % n)
    return (time.perf_counter() - t0) / steps
run(D, False)
ms_hbm = run(steps, False) * 1e3
ms_host = run(steps, True) * 1e3
frames[0].encode(origins, assembling_shard=0)
data = bytes(frames[0].read())
reruns = sum(m.context_overflow_reruns(d) for m in frames for d in range(n))
print("RESULT " + json.dumps({"ms_per_frame": round(ms_hbm, 3), "ms_per_frame_file_to_pinned_host": round(ms_host, 3), "frames_in_flight": D,
                              "shards": n, "devices": devices, "steps": steps, "frame_bytes": len(data), "frame_md5": hashlib.md5(data).hexdigest(),
                              "overflow_reruns": reruns}))
for m in frames:
    m.close()
"""


def shard_inprocess_leg(ndev, size=16384, steps=40):
    """configs[3] through ONE C call per frame in ONE process (include/hydrium_amd.h hydamd_encode_image_multi,
    csrc/host/multi.c): the pixels already in HBM on every device, LF groups dealt in raster runs, floors by peer read,
    every shard's blob a view, the file assembled on a rotating shard's GPU from peer reads — hyd_send_tile's
    multi-device closing stage without the uploads, no RCCL.  One GPU: the list 0,0,0,0 (four contexts, every
    cross-context step, only the xGMI hop missing); at --gpus N rank 0's subprocess drives all N devices."""
    import subprocess

    env = dict(os.environ, PYTHONPATH=ROOT)
    for k in ("HYDAMD_DEVICE", "HYDAMD_DEVICES", "RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-c", _INPROCESS_CLIENT, str(ndev), str(size), str(steps)], capture_output=True, text=True, env=env, timeout=300)
    line = next((l for l in r.stdout.splitlines() if l.startswith("RESULT ")), None)
    if r.returncode or not line:
        return {"error": (r.stderr or r.stdout)[-400:]}
    out = json.loads(line[7:])
    out["aliased"] = ndev <= 1
    out["Mpixel/s"] = round(size * size / out["ms_per_frame"] / 1e3, 1)
    out["workload"] = (f"one {size}x{size} RGB8 'photo' frame per step (BASELINE configs[3]), device-resident on every device, "
                       "hydamd_encode_image_multi + hydamd_multi_result per frame, the finished file left in the assembling device's HBM "
                       "(ms_per_frame) or copied into pinned host memory (ms_per_frame_file_to_pinned_host); wall clock over `steps` frames")
    return out


def main():
    args = parse()
    rc = launch_ranks(args)
    if rc is not None:
        sys.exit(rc)
    if args.dry_run_launch:
        return dry_run_launch(args)
    if args.mode == "shard":
        return run_shard(args)
    if args.mode == "batch":
        return run_batch(args)
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    exchange = args.exchange  # per-frame blob export + RCCL gather (also in a world of one, as a self-test)
    # (HYDAMD_BENCH_FORCE_PG=1: a lone rank forms the RCCL group too — what does the group's mere presence, its streams and
    # their hardware queues, cost the loop?  profiles/r06_pg_presence.txt)
    use_dist = world > 1 or exchange or os.environ.get("HYDAMD_BENCH_FORCE_PG") == "1"  # a process group exists: barriers and the slowest rank's clock
    FPL = 1 if exchange else max(1, args.frames_per_launch)  # frames per launch group (a batch is not exported as one blob)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP extension has no CPU fallback")
    torch.cuda.set_device(local)
    os.environ["HYDAMD_DEVICE"] = str(local)  # the drop-in API legs of this process encode on this rank's GPU
    placement = None
    if not args.no_bind:
        from hydrium_amd import placement as _pl

        placement = _pl.bind_near_gpu(local)  # as a deployment would (numactl): uploads run 20 % slower from the other socket
    os.environ.setdefault("HYDAMD_CONTEXT_CACHE", str(max(4, args.threads)))
    if use_dist:
        for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533"), ("RANK", "0"), ("WORLD_SIZE", "1")):
            os.environ.setdefault(k, v)  # only missing when --exchange is used without torchrun
        # before the contexts exist: HIP hands hardware queues to streams in creation order, and RCCL's streams created
        # after sixteen contexts' land on queues the contexts use (measured in a world of one: 81 instead of 121 Gpixel/s)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        rccl_ranks = check_world(args, dist, torch.device("cuda", local))
    else:
        rccl_ranks = 1
    from hydrium_amd import api, device, sharding, synth

    W = H = args.size
    # each rank's slab is a window of one big synthetic picture, slabs side by side in raster order
    gx = sharding.slab_grid(world)[0]
    img = synth.make_image(args.kind, W, H, args.depth, x0=(rank % gx) * W, y0=(rank // gx) * H,
                           device=torch.device("cuda", local))
    lfg = (-(-W // 2048)) * (-(-H // 2048))
    ctxs = [device.DeviceContext(local, lfg * FPL, 0) for _ in range(max(1, args.streams))]
    # every frame a context holds is a picture of its own — windows of the synthetic image stacked below the ranks' slabs,
    # 0.4 GB each of 288 — so that no cache can serve one context's pixels to another (round 4 coded one tensor everywhere).
    # Context 0's first frame is `img`, the picture the single-frame, API and CPU legs code.
    gy = -(-world // gx)
    pictures = [[img if (k, f) == (0, 0) else
                 synth.make_image(args.kind, W, H, args.depth, x0=(rank % gx) * W, y0=((1 + k * FPL + f) * gy + rank // gx) * H,
                                  device=torch.device("cuda", local)) for f in range(FPL)] for k in range(len(ctxs))]
    torch.cuda.synchronize()
    for c in ctxs:
        c.set_rans_waves(args.rans_waves)
        # throughput loop: the LF coder runs at the end of each context's own stream (mode 2), so that
        # 12 frames in flight stay within the 16 hardware queues; the latency leg uses the side stream
        c.set_lf_coder(2 if args.lf_coder == "on" else 0)

    # N > 1: every frame's results leave each GPU as ONE blob (hydamd_export_frame: tables, section sizes,
    # coded LF streams, packed HF sections) built by a kernel at the end of the frame's stream, and one
    # RCCL gather brings the blobs to the assembling rank — no host synchronisation anywhere: the gather is
    # issued asynchronously under the context's stream and the context's NEXT frame waits for it on the device.
    ext = [torch.cuda.ExternalStream(c.get_stream()) for c in ctxs]
    # A RCCL operation costs most of a millisecond end to end whatever it moves and the operations of a
    # process group run one after another: one gather per frame would cap the rate near 1.1 frames/ms
    # (measured, world of one: 74 instead of 108 Gpixel/s).  So the blobs of `per` consecutive frames
    # (contexts) sit in one tensor and travel together.
    per = max(1, min(args.gather_every, len(ctxs)))
    while len(ctxs) % per:
        per -= 1
    ngroups = len(ctxs) // per
    # two blob buffers per group, used alternately: a context's next frame does not wait for the gather of its previous one
    # (queued behind the other contexts' gathers, it ended after that frame should have started) but for the one before
    xstate = {"cap": 0, "big": [[None, None] for _ in range(ngroups)], "rows": [None] * ngroups,
              "work": [[None, None] for _ in range(ngroups)]}
    xt = [0.0, 0.0]  # host seconds spent issuing the export + gather
    xstream = torch.cuda.Stream() if exchange and per > 1 else None

    def step(i):
        """one launch group (FPL frames) on context i mod S"""
        k = i % len(ctxs)
        ctx = ctxs[k]
        if not exchange:
            if FPL > 1:
                ctx.encode_image_batch(pictures[k])
            else:
                ctx.encode_image_tensor(pictures[k][0])
            return ctx
        j = k // per
        half = (i // len(ctxs)) & 1
        with torch.cuda.stream(ext[k]):
            if xstate["work"][j][half] is not None:
                xstate["work"][j][half].wait()  # device-side: this stream waits until the gather that last read this buffer is done
            ctx.encode_image_tensor(pictures[k][0])
            t_b = time.perf_counter()
            ctx.export_frame(lfg, xstate["big"][j][half][k % per])
            xt[1] += time.perf_counter() - t_b
        if k % per == per - 1:  # the group's last frame is queued: one collective for all of its blobs
            t_b = time.perf_counter()
            if per > 1:
                for q in range(j * per, j * per + per):
                    xstream.wait_stream(ext[q])
            # one frame per collective: issued under the context's own stream (RCCL's stream waits for it through an event)
            # the assembling rank rotates with the collective (as in --mode shard): every rank receives world blobs every
            # world-th time instead of rank 0 receiving all of them — inbound bytes per rank = its outbound bytes, and the
            # max-over-ranks time is not one rank's
            root = (i // per) % world
            with torch.cuda.stream(xstream if per > 1 else ext[k]):
                xstate["work"][j][half] = dist.gather(xstate["big"][j][half].view(-1),
                                                      gather_list=xstate["rows"][j] if rank == root else None, dst=root,
                                                      async_op=True)
            xt[1] += time.perf_counter() - t_b
        return ctx

    def drain():
        for j, pair in enumerate(xstate["work"]):
            for half, w in enumerate(pair):
                if w is not None:
                    for q in range(j * per, j * per + per):
                        with torch.cuda.stream(ext[q]):
                            w.wait()
                    pair[half] = None

    # initialisation, not measurement: every context codes one frame once so that its freshly
    # allocated buffers have been touched before anything is timed; then the W warm-up steps
    for k, c in enumerate(ctxs):
        step_init = c.encode_image_batch(pictures[k]) if FPL > 1 else c.encode_image_tensor(pictures[k][0])
    for c in ctxs:
        c.sync()
    if exchange:
        # blob size every rank sends: 1.25 x the largest blob of this first frame over all ranks
        probe = torch.zeros(ctxs[0].blob_bound(lfg), dtype=torch.uint8, device=img.device)
        with torch.cuda.stream(ext[0]):
            ctxs[0].export_frame(lfg, probe)
        ctxs[0].sync()
        mine_total = int(device.blob_header(probe[:64].cpu().numpy().tobytes())["total_bytes"])
        capt = torch.tensor([mine_total], dtype=torch.int64, device=img.device)
        dist.all_reduce(capt, op=dist.ReduceOp.MAX)
        xstate["cap"] = (int(int(capt.item()) * 1.25) + 65536 + 15) & ~15
        del probe
        for j in range(ngroups):
            xstate["big"][j] = [torch.zeros((per, xstate["cap"]), dtype=torch.uint8, device=img.device) for _ in range(2)]
            # every rank takes its turn as the root
            xstate["rows"][j] = [torch.empty(per * xstate["cap"], dtype=torch.uint8, device=img.device) for _ in range(world)]
    for i in range(args.warmup):
        step(i)
    drain()
    for c in ctxs:
        c.sync()

    # ---- the timed interval ----
    # A frame takes several milliseconds from first kernel to last while a new one completes every
    # fraction of one: K frames bracketed by two synchronisations would measure fill and drain of the
    # pipeline, not its rate (at K = 20 the drain alone was a quarter of the region).  So the pipeline is
    # primed with four frames per context, the timed frames follow, then one more frame per context
    # keeps it full while the timed ones finish.  Every frame leaves a HIP event at the end of its
    # context's stream; the interval runs from the completion of the last priming frames to the
    # completion of the last timed frames — frame completions at the pipeline's own rate.  The whole
    # sequence still sits between barrier + synchronize on both sides (`wall` below reports it).
    # Frames of the S streams complete in bursts, so an interval shorter than two periods of every stream
    # moves by several per cent with where it happens to start (round 2: 136-146 at --steps 20, 128-133 at 120 on the
    # same box): at least 2 S frames are timed whatever --steps says, and the rate is reported per frame.
    S = len(ctxs)
    # at least eight periods of every stream per window (128 frames, ~65 ms), three windows: the contract line is the MEDIAN
    # window, the spread is reported.  The event timers of hydamd_profile stay OFF in these windows (two event records per
    # kernel lengthen every frame's stay in its stream: -4 % measured); the co-residency durations under `kernels` come from
    # a fourth, short, profiled window that is not part of the rate.
    K = max(args.steps, 8 * S * FPL)   # frames per window
    K = -(-K // FPL) * FPL              # a whole number of launch groups
    KG = K // FPL                       # launch groups per window
    WINDOWS = 3

    def timed_run(K, step=step, windows=1, nprime=None):
        """`windows` consecutive windows of K launch groups each (FPL frames per group for the default `step`), timed by
        events, in ONE continuous run behind a priming phase.  The GPU's first ~150 ms under this all-VALU load run 10-15 %
        faster than what follows (clocks settle: the transform kernel alone, back to back on 32 streams, does 216-241
        Gpixel/s for two chunks of frames and 198-202 for the next seconds: profiles/r04_pipeline_bounds.txt), so the priming
        phase is 10 launch groups per stream (>= 0.2 s) and what is timed is the sustained rate."""
        nprime = 10 * S if nprime is None else nprime
        ncool = S

        def mark(i):
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(ext[i % S])
            return ev

        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        seq = 0
        evs = []
        for _ in range(nprime + windows * K):
            step(seq)
            evs.append(mark(seq))
            seq += 1
        for _ in range(ncool):
            step(seq)
            seq += 1
        drain()
        for c in ctxs:
            c.sync()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        wall = time.perf_counter() - t0
        # all events on one clock: offsets from the first priming group's completion; a window boundary is the mean
        # completion time of the S launch groups in front of it — boundaries exactly K groups apart
        base = evs[0]
        done = [base.elapsed_time(e) for e in evs]
        w = min(S, K, nprime)
        edge = [sum(done[nprime + j * K - w:nprime + j * K]) / w for j in range(windows + 1)]
        dts = [(edge[j + 1] - edge[j]) * 1e-3 for j in range(windows)]
        # every stream's own period inside the timed part (it completes one launch group per S group periods)
        periods = []
        for k in range(S):
            mine = [done[i] for i in range(nprime, nprime + windows * K) if i % S == k]
            if len(mine) >= 2:
                periods.append((mine[-1] - mine[0]) / (len(mine) - 1) / S / FPL)
        return dict(dt=sorted(dts)[windows // 2], dts=dts, wall=wall, frames=seq, periods=periods)

    run = timed_run(KG, windows=WINDOWS)
    if use_dist:  # every window's time is the slowest rank's
        t = torch.tensor(run["dts"] + [run["wall"]], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        row = t.tolist()
        run["dts"], run["wall"] = row[:-1], row[-1]
        run["dt"] = sorted(run["dts"])[WINDOWS // 2]
    dt, wall, total_frames = run["dt"], run["wall"], run["frames"] * FPL
    window_rates = [round(world * W * H * K / x / 1e6, 1) for x in run["dts"]]
    # the first 100 ms after idle, for the record: a short window right behind a short priming phase (what rounds 1-3 timed)
    burst = timed_run(2 * S, nprime=2 * S)
    for c in ctxs:
        c.profile(True)
    timed_run(2 * S, nprime=2 * S)
    groups_profiled = FPL

    # per-kernel durations from HIP events recorded on the kernels' own streams during the timed region
    kern = {}
    for c in ctxs:
        for name, (ms, n) in c.profile_read().items():
            a = kern.setdefault(name, [0.0, 0])
            a[0] += ms
            a[1] += n
        c.profile(False)
    payload_bytes = ctxs[0].payload_size() // FPL
    symbols = sum(int(ctxs[0].read_symbol_counts(s).sum()) for s in range(lfg))

    # What the timed contexts hold.  Every context still holds its last timed launch group — the very launch shape that was
    # timed (S in flight, FPL frames per group, lane-form entropy stage, in-stream LF coder): each frame's packed HF sections
    # and coded LF streams are hashed.  The same frame is then coded alone on a context of its own, hashed the same way,
    # exported and put together on the device (hydamd_assembler_*) into the FILE, whose MD5 is compared below with the
    # drop-in API's file and with the CPU reference's.
    timed_files = None
    if rank == 0 and args.lf_coder == "on" and W * H > 65536:
        def frame_digests(c, frames):
            c.sync()
            pay = c.read_payload()
            lfr = c.read_lf_streams(lfg * frames)
            lfp = c.read_lf_payload()
            out, off = [], 0
            for f in range(frames):
                n = sum(int(((c.read_sections(f * lfg + sl)[0] + 7) // 8).sum()) for sl in range(lfg))
                h = hashlib.md5(pay[off:off + n])
                off += n
                for r in lfr[f * lfg:(f + 1) * lfg]:
                    o, nb = int(r["offset"]), (int(r["bit_count"]) + 7) // 8
                    h.update(bytes(r["lengths"]))
                    h.update(bytes(lfp[o:o + nb]))
                out.append(h.hexdigest())
            return out

        held = [d for c in ctxs for d in frame_digests(c, FPL)]
        md = api.HYDImageMetadata(W, H, 0, -1, -1)
        with device.DeviceContext(local, lfg, 0) as v, device.Assembler(local) as asm:
            v.set_rans_waves(args.rans_waves)
            v.set_lf_coder(2)
            alone = []
            for k in range(len(ctxs)):  # every picture coded alone, in the order `held` lists them; `img` (context 0, frame 0) last
                for f in range(FPL):
                    if (k, f) != (0, 0):
                        v.encode_image_tensor(pictures[k][f])
                        alone.append(frame_digests(v, 1)[0])
            v.encode_image_tensor(img)
            alone.insert(0, frame_digests(v, 1)[0])
            asm.plan(md, [list(range(lfg))])
            out_buf = torch.empty(v.blob_bound(lfg) + (1 << 20), dtype=torch.uint8, device=img.device)
            vs = torch.cuda.ExternalStream(v.get_stream())
            with torch.cuda.stream(vs):
                ptr, cap = v.export_frame_owned(lfg)
                asm.run([ptr], [cap], out_buf.data_ptr(), out_buf.numel(), vs.cuda_stream)
            v.sync()
            file_md5 = hashlib.md5(out_buf[:asm.result()].cpu().numpy()).hexdigest()
            del out_buf
        timed_files = {"contexts": len(ctxs), "frames_per_launch_group": FPL, "frames_hashed": len(held),
                       "distinct_pictures": len(set(alone)),
                       "sections_and_lf_streams_equal_the_frame_coded_alone": held == alone,
                       "md5": file_md5, "md5_is": "context 0's first picture coded alone, exported and assembled on the device"}

    # single-frame latency leg: one stream, one wave per group (the lowest-latency entropy form),
    # each frame synchronised before the next starts; kernels run alone, so these are also the
    # un-overlapped kernel durations
    lat = None
    lat5 = None
    if world == 1:
        c0 = ctxs[0]
        # the throughput form's own un-overlapped durations
        c0.set_rans_waves(max(5, args.rans_waves))
        c0.set_lf_coder(2 if args.lf_coder == "on" else 0)  # in-stream: no kernel of the frame overlaps another
        c0.encode_image_tensor(img)
        c0.sync()
        c0.profile(True)
        tl = time.perf_counter()
        for _ in range(5):
            c0.encode_image_tensor(img)
            c0.sync()
        tl = (time.perf_counter() - tl) / 5
        lat5 = {"ms_per_frame": round(tl * 1e3, 4),
                "kernel_avg_ms": {k: round(ms / max(n, 1), 4) for k, (ms, n) in c0.profile_read().items()}}
        c0.profile(False)
        c0.set_rans_waves(4)
        if args.lf_coder == "on":
            c0.set_lf_coder(1)
        c0.encode_image_tensor(img)
        c0.sync()
        c0.profile(True)
        reps = 5
        tl = time.perf_counter()
        for _ in range(reps):
            c0.encode_image_tensor(img)
            c0.sync()
        tl = (time.perf_counter() - tl) / reps
        lk = {k: round(ms / max(n, 1), 4) for k, (ms, n) in c0.profile_read().items()}
        c0.profile(False)
        c0.set_rans_waves(args.rans_waves)
        if args.lf_coder == "on":
            c0.set_lf_coder(2)
        lat = {"ms_per_frame": round(tl * 1e3, 4), "Mpixel/s": round(W * H / tl / 1e6, 1), "kernel_avg_ms": lk,
               "note": "one stream, one frame at a time, rANS form 4 (one wave per group); kernels not overlapped"}

    # reference leg: the same loop, timed the same way, with the LF-group coder switched off (SURVEY.md 8(d)(ii)'s
    # narrower definition: device-resident input -> HF group sections only)
    hf_only = None
    if world == 1 and args.lf_coder == "on" and not args.no_legs:
        for c in ctxs:
            c.set_lf_coder(False)
        r2 = timed_run(4 * S)
        hf_only = {"Mpixel/s": round(W * H * 4 * S * FPL / r2["dt"] / 1e6, 1), "frames": 4 * S * FPL,
                   "note": "same loop and same event-window timing, LF coder off: HF group sections only, LF ints left for a host coder"}
        for c in ctxs:
            c.set_lf_coder(2)

    # the same loop with ONE frame per launch group (what rounds 1-3 timed)
    one_per_group = None
    if world == 1 and FPL > 1 and not args.no_legs:
        r1 = timed_run(8 * S, lambda i: ctxs[i % S].encode_image_tensor(pictures[i % S][0]))
        one_per_group = {"Mpixel/s": round(W * H * 8 * S / r1["dt"] / 1e6, 1), "ms_per_step": round(r1["dt"] / (8 * S) * 1e3, 4), "frames": 8 * S,
                         "note": "same loop, contexts and timing, hydamd_encode_image: every frame a launch group of its own"}

    # and with the frame FINISHED in the loop: every step also exports the context's results and puts the codestream together
    # on the device (hydamd_assembler_*), so that a step ends with the complete .jxl file in HBM instead of its sections
    whole_file = None
    if world == 1 and args.lf_coder == "on" and not args.no_legs and W * H > 65536:
        md = api.HYDImageMetadata(W, H, 0, -1, -1)
        asms = [device.Assembler(local) for _ in ctxs]
        cap = ctxs[0].blob_bound(lfg)
        outs_w = [torch.empty(cap + (1 << 20), dtype=torch.uint8, device=img.device) for _ in ctxs]
        for a in asms:
            a.plan(md, [list(range(lfg))])

        def step_file(i):
            k = i % S
            with torch.cuda.stream(ext[k]):
                ctxs[k].encode_image_tensor(pictures[k][0])
                ptr, n = ctxs[k].export_frame_owned(lfg)  # a view: records only, the sections stay in the context's buffers
                asms[k].run([ptr], [n], outs_w[k].data_ptr(), outs_w[k].numel(), ext[k].cuda_stream)

        r3 = timed_run(4 * S, step_file)
        sizes = {a.result() for a in asms}
        digest = hashlib.md5(outs_w[0][:asms[0].result()].cpu().numpy()).hexdigest()
        whole_file = {"Mpixel/s": round(W * H * 4 * S / r3["dt"] / 1e6, 1), "ms_per_step": round(r3["dt"] / (4 * S) * 1e3, 4),
                      "frames": 4 * S, "file_bytes": sorted(sizes), "md5": digest,
                      "note": "same loop and timing; each step also runs hydamd_export_frame_owned and the device-side assembler: "
                              "the finished codestream (file header, frame header, TOC, every section) is in HBM when the step ends"}
        for a in asms:
            a.close()
        del outs_w

    out = None
    if rank == 0:
        bytes_in = W * H * 3 * (args.depth // 8)
        kernels = {k: {"avg_ms": round(v[0] / max(v[1], 1), 4), "launches": v[1],
                       "algorithmic_GBs": round(bytes_in * groups_profiled / (v[0] / max(v[1], 1) * 1e-3) / 1e9, 1) if v[0] else None}
                   for k, v in kern.items()}
        # dominant kernel = the one that lasts longest with the GPU to itself (single-frame leg of this same run, the
        # loop's form): the serial rANS chains.  Inside the interval per-launch durations are co-residency figures (a
        # launch there shares the GPU with the other streams' kernels and lasts several frame periods), and by their sum
        # the chains and the transform kernel trade places from run to run; roofline_transform_kernel is always K1's.
        alone = (lat5 or {}).get("kernel_avg_ms", {}) if args.rans_waves >= 5 else (lat or {}).get("kernel_avg_ms", {})
        dom = max(alone, key=lambda k: alone[k]) if alone else max(kern, key=lambda k: kern[k][0])
        dom_ms = alone.get(dom) or kern[dom][0] / max(kern[dom][1], 1)
        achieved = bytes_in / (dom_ms * 1e-3) / 1e9
        k1_ms = alone.get("transform_tokenize")
        traffic = None
        valu = None
        ceiling = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as f:
                tj = json.load(f)
            form = f":form{args.rans_waves}" if dom == "rans_encode" else ""
            traffic = tj.get(f"{dom}{form}:{W}x{H}:u{args.depth}:{args.kind}")
            vk = tj.get(f"valu:transform_tokenize:{W}x{H}:u{args.depth}:{args.kind}")
            if vk and k1_ms:
                # lane-operations the transform kernel issues (rocprofv3 SQ_INSTS_VALU x 64, profiles/) over
                # its un-overlapped duration, against the issue peak scripts/ubench/valu_rate measures and against
                # the nominal one (4 SIMD x 32 lanes x 256 CUs x 2.4 GHz)
                ops = vk["valu_wave_instructions"] * 64
                rate = ops / (k1_ms * 1e-3) / 1e12
                nominal = vk.get("nominal_peak_T_lane_ops_s", 78.6)
                valu = {"kernel": "transform_tokenize", "lane_ops_per_pixel": round(ops / (W * H), 1),
                        "achieved": round(rate, 2), "peak": vk["peak_T_lane_ops_s"], "unit": "T lane-ops/s",
                        "frac": round(rate / vk["peak_T_lane_ops_s"], 3), "peak_source": vk.get("peak_source"),
                        "peak_nominal": nominal, "frac_of_nominal": round(rate / nominal, 3),
                        "exact_arithmetic_floor_lane_ops_per_pixel": vk.get("floor_lane_ops_per_pixel")}
                floor = vk.get("floor_lane_ops_per_pixel")
                if floor:
                    # the reference's non-fused arithmetic cannot be shared, re-associated or fused without changing
                    # bytes: even at the best issue rate measured the transform kernel needs floor x pixels / peak
                    t_floor = floor * W * H / (vk["peak_T_lane_ops_s"] * 1e12)
                    ceiling = {"frac_of_hbm_read_roofline": round(bytes_in / t_floor / (HBM_PEAK_GBS * 1e9), 3),
                               "Mpixel/s": round(W * H / t_floor / 1e6, 0), "ms_per_frame": round(t_floor * 1e3, 4),
                               "basis": f"{floor} lane-ops per pixel of bit-exact arithmetic (DESIGN.md 3) at {vk['peak_T_lane_ops_s']} T lane-ops/s, "
                                        "the best VALU issue rate measured on this chip; the guard-banded fast DCT that could lower it "
                                        "was measured and dropped (profiles/r03_guardband.txt)"}
        per_stream = run["periods"]
        out = {
            "metric": "Mpixel/s encode (8K RGB, default q)",
            "value": round(world * W * H * K / dt / 1e6, 1),
            "unit": "Mpixel/s",
            "n_gpus": world, "rccl_ranks": rccl_ranks, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / K * 1e3, 4),
            "value_is": "the SUSTAINED rate: median of three consecutive windows (timing.timed_frames frames each) of one continuous run behind ten launch groups per stream of priming "
                        "(it equals the barrier-to-barrier rate of the whole run, timing.Mpixel/s_wall, within a few per cent)",
            "value_by_the_method_of_rounds_1_to_3": round(world * W * H * 2 * S * FPL / burst["dt"] / 1e6, 1),
            "value_by_the_method_of_rounds_1_to_3_is": "a short event window right behind a short priming phase, from an idle GPU (BENCH_r03: 136 941 by this "
                                                       "method); the GPU's first 150 ms under this load run 10-15 % faster than what follows (DESIGN.md 4)",
            "timing": {"method": "HIP events at the end of each frame's stream: completion of the last priming frames -> "
                                 "completion of the last timed frames, pipeline primed before and kept full behind; "
                                 "one continuous run: ten launch groups per stream of priming (the GPU's first 150 ms under this load run 10-15 % fast), "
                                 "then three consecutive windows of at least eight launch groups per stream (timed_frames frames each); the rate is "
                                 "per frame of the median window = the SUSTAINED rate; no event timers inside the windows",
                       "timed_frames": K, "windows": WINDOWS, "Mpixel/s_each_window": window_rates,
                       "Mpixel/s_first_100ms_after_idle": round(world * W * H * 2 * S * FPL / burst["dt"] / 1e6, 1),
                       "first_100ms_note": "a window of two launch groups per stream right behind two of priming, from an idle GPU: the method of "
                                           "rounds 1-3 (round 3's driver line, 136.9 Gpixel/s, was measured this way); `value` is the sustained rate",
                       "spread_pct": round(100.0 * (max(window_rates) - min(window_rates)) / (sum(window_rates) / len(window_rates)), 2),
                       "value_is": "the median window",
                       "per_stream_ms_per_step": ({"min": round(min(per_stream), 4), "mean": round(sum(per_stream) / len(per_stream), 4),
                                                   "max": round(max(per_stream), 4)} if per_stream else None),
                       "transform_kernel_alone_ms_over_ms_per_step": round(k1_ms / (dt / K * 1e3), 3) if k1_ms else None,
                       "wall_ms_incl_fill_and_drain": round(wall * 1e3, 3), "frames_in_wall": total_frames,
                       "Mpixel/s_wall": round(world * W * H * total_frames / wall / 1e6, 1)},
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "host_placement": placement or "unbound",
            "config": {"workload": f"{W}x{H} RGB{args.depth} '{args.kind}' frame per GPU (BASELINE configs[2]); "
                                   "hot path device-resident RGB -> packed HF group sections "
                                   "(XYB, DCT, quantise, tokenise, ANS tables, rANS, pack)" +
                                   (f", {FPL} independent frames per launch group" if FPL > 1 else "") +
                                   (" + prefix-coded LF coefficient streams" if args.lf_coder == "on" else ""),
                       "lf_coder": "gpu, in-stream" if args.lf_coder == "on" else "off",
                       "groups": lfg * 64 if W % 2048 == 0 and H % 2048 == 0 else None, "lf_groups": lfg,
                       "streams": len(ctxs), "frames_per_launch_group": FPL, "rans_groups_per_workgroup": args.rans_waves, "parallelism": f"{world} x (one frame per GPU)" +
                                                            (f", one RCCL gather of result blobs per {per} frames, root rotating over the ranks" if exchange else ", no collective (independent frames)")},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "algorithmic_bytes_per_launch": bytes_in, "avg_launch_ms": round(dom_ms, 4),
                         "duration_source": "this kernel alone (single-frame leg of this run), not overlapped with other streams"},
            "roofline_transform_kernel": ({"bound": "hbm", "kernel": "transform_tokenize",
                                           "achieved": round(bytes_in / (k1_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                                           "unit": "GB/s", "frac": round(bytes_in / (k1_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                           "avg_launch_ms": k1_ms} if k1_ms else None),
            "valu_roofline": valu,
            "hbm_ceiling_under_exact_arithmetic": ceiling,
            "kernels": kernels,
            "kernels_note": "per-launch durations (a launch covers frames_per_launch_group frames) in a separate short profiled window of the same loop, where the streams' kernels overlap each other",
            "exchange": ({"blob_capacity_bytes": xstate["cap"], "host_ms_per_step_issuing_export_and_gather":
                          round(xt[1] / max(total_frames + args.warmup, 1) * 1e3, 4),
                          "frames_per_collective": per,
                          "note": "one hydamd_export_frame kernel per frame, one asynchronous RCCL gather per group of frames, no host synchronisation"}
                         if exchange else None),
            "timed_contexts_as_files": timed_files,
            "single_frame": lat,
            "single_frame_form5": lat5,  # the loop's own lane-per-group form, un-overlapped
            "hf_sections_only": hf_only,
            "one_frame_per_launch_group": one_per_group,
            "finished_file_per_step": whole_file,
            "symbols_per_pixel": round(symbols / (W * H), 4),
            "section_bytes": payload_bytes,
            "hbm_read_roofline_Mpx_s": round(HBM_PEAK_GBS * 1e9 / (3 * args.depth // 8) / 1e6, 0),
            "frac_of_hbm_read_roofline": round((W * H * K / dt) / (HBM_PEAK_GBS * 1e9 / (3 * args.depth // 8)), 5),
        }
    host_img = None
    if rank == 0 and world == 1 and not (args.no_cpu_baseline and args.no_api):
        arr = img.cpu().numpy()
        host_img = np.ascontiguousarray(arr.view(np.uint16) if args.depth == 16 else arr)

    # the frame-mode contexts are done: give their memory and streams back before the other workloads run
    drain()
    xstate["big"] = xstate["rows"] = None
    import gc

    gc.collect()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.empty_cache()  # blocks torch handed out under the contexts' streams go back before the streams do
    for c in ctxs:
        c.close()
    del img
    pictures = None
    gc.collect()
    torch.cuda.empty_cache()

    # ---- the other two BASELINE workloads, a few hundred milliseconds each, in the same line: configs[3] (one 16K frame
    # sharded over the GPUs, assembled on the device) and configs[4] (a batch of 4K frames through the drop-in API) ----
    legs = {}
    if not args.no_legs and world == 1:  # (N > 1: as jobs of their own once the group is gone, see child_job)
        def shard_16k():
            if not dist.is_initialized():  # a lone rank needs the process group for this leg only: it comes last
                for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533"), ("RANK", "0"), ("WORLD_SIZE", "1")):
                    os.environ.setdefault(k, v)
                dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            return shard_leg(args, 40, 4, 16384, assemble=args.assemble)

        # configs[4] is a batch of 64 frames: 18 ms / 45 ms of work, too short to time; eight batches back to back
        for name, fn in (("batch_4k_device", lambda: batch_device_leg(args, 8 * args.frames)),
                         ("batch_4k", lambda: batch_leg(args, 8 * args.frames, args.threads)), ("shard_16k", shard_16k)):
            try:
                t_leg = time.perf_counter()
                r = fn()
                if r is not None:
                    r["leg_wall_s"] = round(time.perf_counter() - t_leg, 2)
                # the drop-in library parks the contexts of destroyed encoders for the next one; their streams keep their
                # places in the runtime's rotation over the hardware queues: the next leg starts from a clean slate
                trim = getattr(api.Library().dll, "hydamd_trim_cache", None)
                if trim is not None:
                    trim.restype = None
                    trim()
                legs[name] = r
            except Exception as exc:  # a leg must not take the headline with it
                legs[name] = {"error": f"{type(exc).__name__}: {exc}"}
    if rank == 0:
        for name, r in legs.items():
            if r is None:
                continue
            out[name] = {k: r[k] for k in LEG_KEYS if k in r}
        if world == 1 and not args.no_api:
            # API end-to-end through the drop-in hyd_send_tile (host pixels: includes PCIe, read-back, assembly)
            lib = api.Library()
            t1 = time.perf_counter()
            import ctypes

            # one output buffer for all frames, large enough for a whole file: one provide/flush/release round per tile
            big = dict(out_buf=(ctypes.c_uint8 * (32 << 20))())
            api.encode_image(lib, host_img, **big)  # first use: creates the device context and its pinned staging
            t_first = time.perf_counter() - t1
            times = []
            for _ in range(5):
                t1 = time.perf_counter()
                data = api.encode_image(lib, host_img, in_place=True, **big)  # the file where a C caller finds it: in its buffer
                times.append(time.perf_counter() - t1)
            data = bytes(data)
            t_api = sorted(times)[len(times) // 2]
            out["api_end_to_end"] = {"Mpixel/s": round(W * H / t_api / 1e6, 1), "ms": round(t_api * 1e3, 1),
                                     "ms_each": [round(x * 1e3, 1) for x in times],
                                     "first_call_ms": round(t_first * 1e3, 1),
                                     "bytes": len(data), "md5": hashlib.md5(data).hexdigest(),
                                     "note": "host-pointer hyd_send_tile path, one-frame mode, one 32 MiB output buffer, median of 5 frames after the "
                                             "first (which also creates the device context); PCIe, frame assembly on the device, one read-back and "
                                             "hyd_flush's copy into the caller's buffer inclusive (until round 4 also a 12 MB Python bytes copy of that buffer: 0.9 ms)"}
            if timed_files:
                timed_files["identical_to_api_file"] = timed_files["md5"] == out["api_end_to_end"]["md5"]
            if whole_file:
                whole_file["identical_to_api_file"] = whole_file["md5"] == out["api_end_to_end"]["md5"]
        if world == 1 and not args.no_content and not args.no_legs and not args.no_api:
            # SURVEY 8(d)'s other two synthetic inputs, each with its own CPU baseline and byte-equality verdict
            out["content"] = {}
            for kind in ("smooth", "noise"):
                try:
                    out["content"][kind] = content_row(args, kind, local)
                except Exception as exc:
                    out["content"][kind] = {"error": f"{type(exc).__name__}: {exc}"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(host_img)
            if "api_end_to_end" in out and "md5" in out["cpu_baseline"]:
                out["api_end_to_end"]["identical_to_cpu_reference"] = out["cpu_baseline"]["md5"] == out["api_end_to_end"]["md5"]
            if timed_files and "md5" in out["cpu_baseline"]:
                timed_files["identical_to_cpu_reference"] = timed_files["md5"] == out["cpu_baseline"]["md5"]
    # tear down first, print last: RCCL writes its version banner to stdout when the process group goes away,
    # and the line the driver parses should be the final one
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    # Rank 0's two multi-device legs run in processes of their own over EVERY GPU of the job.  They come after the process group is
    # gone: the other ranks are not held in a collective meanwhile (a leg that hangs on hardware it has never met would run the
    # group's watchdog out and take the headline with it), and at --gpus N their GPUs are idle when rank 0's subprocess takes them.
    if rank == 0:
        if world > 1 and not args.no_legs:
            # configs[3] and configs[4] over all N GPUs, each a job of its own (in a world of one they ran in-process above)
            for name, argv in (("shard_16k", ["--mode", "shard", "--steps", "40", "--warmup", "4"]),
                               ("batch_4k", ["--mode", "batch", "--frames", str(8 * args.frames), "--threads", str(args.threads)])):
                t_leg = time.perf_counter()
                r = child_job(world, argv + (["--no-bind"] if args.no_bind else []))
                r["leg_wall_s"] = round(time.perf_counter() - t_leg, 2)
                out[name] = {k: r[k] for k in LEG_KEYS if k in r}
                out[name]["ran_as"] = f"a job of its own after the frame loop's process group was gone: bench.py --gpus {world} " + " ".join(argv)
        if not args.no_legs and not args.no_api:
            # the C library's own multi-device scheduler, over every GPU of the job (one GPU: an aliased list), once, on rank 0
            try:
                out["api_multi_device"] = api_multi_device_leg(world)
                if "shard_16k" in out and "frame_md5" in out["shard_16k"] and "md5" in out["api_multi_device"]:
                    out["api_multi_device"]["same_file_as_shard_16k"] = out["api_multi_device"]["md5"] == out["shard_16k"]["frame_md5"]
            except Exception as exc:
                out["api_multi_device"] = {"error": f"{type(exc).__name__}: {exc}"}
        if not args.no_legs:
            # the same frame as shard_16k through the C library's in-process composition (no RCCL, no host pixels)
            try:
                out["shard_16k_inprocess"] = shard_inprocess_leg(world)
                if "shard_16k" in out and "frame_md5" in out["shard_16k"] and "frame_md5" in out["shard_16k_inprocess"]:
                    out["shard_16k_inprocess"]["same_file_as_shard_16k"] = out["shard_16k_inprocess"]["frame_md5"] == out["shard_16k"]["frame_md5"]
            except Exception as exc:
                out["shard_16k_inprocess"] = {"error": f"{type(exc).__name__}: {exc}"}
        if "cpu_baseline" in out:  # (the reported baseline stays the line's last object)
            out["cpu_baseline"] = out.pop("cpu_baseline")
        emit(out)


if __name__ == "__main__":
    main()
