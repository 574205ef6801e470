"""GPU: the multi-device tile scheduler INSIDE libhydrium.so.0 (SURVEY 8(e) behind the C boundary).

hyd_send_tile deals a one-frame image's LF groups to the devices HYDAMD_DEVICES names — runs of consecutive tiles, one
context per device, the running alphabet maximum exchanged by a peer read, the frame assembled on the first shard's GPU
from every shard's blob read in place (csrc/host/encoder.c finish_frame_multi; reference libhydrium.c:172-203,
encoder.c:928-957).  A box with one GPU runs the very same code with the device list ALIASED (0,0,0,0: four contexts,
four streams, every cross-context step taken — only the xGMI hop is missing), which is what these tests do; the device
list is read once per process, so every case runs in a process of its own.  On more than one physical GPU this path is
unmeasured (no such box in this environment)."""
import os
import subprocess
import sys

import pytest

from conftest import has_gpu, reference_expected
from hydrium_amd import build as hbuild

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_ENCODE = r"""
import hashlib, json, sys
import numpy as np
import torch
from hydrium_amd import api, synth
from oracle import refprobe
a = json.loads(sys.argv[1])
kind, w, h, depth = a["case"]
if depth == 32:
    img = synth.make_image_f32(kind, w, h)
else:                                   # generated on the GPU: a 16384 x 16384 picture takes the CPU most of a minute
    t = synth.make_image(kind, w, h, depth, device="cuda")
    torch.cuda.synchronize()
    img = t.cpu().numpy()
    img = np.ascontiguousarray(img.view(np.uint16) if depth == 16 else img)
    del t
if a.get("nan"):
    img = img.copy(); img[h // 2, w // 2, 1] = np.nan
tiles = None
if a.get("order") == "reversed":
    ntx, nty = -(-w // 2048), -(-h // 2048)
    tiles = [(tx, ty) for ty in range(nty) for tx in range(ntx)][::-1]
lib = api.Library()
def one(tag):
    try:
        data = api.encode_image(lib, img, order=tiles)
        print(tag, len(data), hashlib.md5(data).hexdigest(), flush=True)
    except api.HydriumError as e:
        print(tag, "ERR", e, flush=True)
if a.get("threads"):                    # encoders side by side, one per thread (ctypes releases the GIL inside the library)
    import threading
    ts = [threading.Thread(target=one, args=("OURS",)) for _ in range(a["threads"])]
    [t.start() for t in ts]; [t.join() for t in ts]
else:
    for k in range(a.get("repeat", 1)):  # one encoder after another in ONE process (the device list, the latches and the pool persist)
        print(f"== image {k}", file=sys.stderr, flush=True)
        one("OURS")
if a.get("reference"):
    ref = api.encode_image(refprobe.reference_library(optimised=True), img, order=tiles)
    print("REF", len(ref), hashlib.md5(ref).hexdigest())
"""


def _run(case, devices, extra_env=None, reference=True, **kw):
    """-> (our result line, the reference's result line for the same picture, stderr)"""
    import json

    if reference:
        assert reference_expected()
    env = dict(os.environ, PYTHONPATH=ROOT, GPU_MAX_HW_QUEUES="22", HYDAMD_DEVICES=devices, HYDAMD_TRACE="1")
    env.pop("HYDAMD_DEVICE", None)
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, "-c", _ENCODE, json.dumps(dict(case=list(case), reference=reference, **kw))],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.strip().splitlines()
    all_ours = [l[5:] for l in lines if l.startswith("OURS ")]
    ref = next((l[4:] for l in lines if l.startswith("REF ")), None)
    if kw.get("repeat") or kw.get("threads"):
        return all_ours, ref, r.stderr
    return all_ours[0], ref, r.stderr


@pytest.mark.parametrize("case,devices", [(("photo", 4296, 4168, 8), "0,0,0,0"), (("photo", 4296, 4168, 8), "0,0"),
                                          (("photo", 16384, 16384, 8), "0,0,0,0"), (("photo", 8192, 6200, 16), "0,0,0")],
                         ids=["4296x4168-4dev", "4296x4168-2dev", "16384x16384-4dev", "8192x6200-u16-3dev"])
def test_frame_dealt_to_an_aliased_device_list_equals_the_reference(case, devices):
    """the acceptance cases: 4296x4168 (9 LF groups, ragged) and 16384x16384 photo (64 LF groups, configs[3])"""
    ours, ref, err = _run(case, devices)
    assert "(shard)" in err, "the multi-device path did not run"
    assert ours == ref


def test_out_of_order_tiles_float_input_and_the_overflow_rerun_across_shards():
    ours, ref, _ = _run(("photo", 4296, 4168, 8), "0,0,0", order="reversed")
    assert ours == ref
    # float samples: growing alphabet from shard to shard (the floor exchange decides log_alphabet_size)
    ours, ref, _ = _run(("photo", 4100, 4100, 32), "0,0,0,0")
    assert ours == ref
    # a shard whose frame outgrows its token arrays reruns it; its blob was stale when the assembler first ran
    ours, ref, _ = _run(("noise", 4296, 4168, 8), "0,0,0,0", {"HYDAMD_TOKEN_CAP": "40000"})
    assert ours == ref
    # ... and with float samples: the shard that reruns had left incomplete alphabet maxima the first time (a group out of
    # token space stops counting), the later shards read their floor from them — they read again and run again (round 6)
    ours, ref, _ = _run(("noise", 4100, 4100, 32), "0,0,0,0", {"HYDAMD_TOKEN_CAP": "40000"})
    assert ours == ref


def test_nan_sample_in_one_shard_is_an_api_error():
    got, _, _ = _run(("photo", 4100, 4100, 32), "0,0,0,0", nan=True, reference=False)
    assert got.startswith("ERR") and "NaN" in got


def test_small_frames_stay_on_one_device_and_encoders_take_devices_in_turn():
    """below HYDAMD_SHARD_MIN_LF_GROUPS (8) a frame is not dealt out: 3840x2160 (4 LF groups, configs[4]) runs on one
    device, and successive encoders take the listed devices in turn"""
    got, ref, err = _run(("photo", 3840, 2160, 8), "0,0,0,0")
    assert "(shard)" not in err
    assert got == ref


def test_c_client_unchanged_runs_on_the_device_list(tmp_path):
    """tests/c/api_client.c, not a line changed, with HYDAMD_DEVICES in its environment"""
    from oracle import refprobe
    from test_c_client import _build

    hbuild.build()
    ours = str(tmp_path / "client_amd")
    _build(ours, os.path.dirname(hbuild.LIB_PATH), os.path.basename(hbuild.LIB_PATH))
    env = dict(os.environ, HYDAMD_DEVICES="0,0,0,0", HYDAMD_TRACE="1")
    env.pop("HYDAMD_DEVICE", None)
    r = subprocess.run([ours, "6200", "4200"], check=True, capture_output=True, text=True, timeout=600, env=env)
    assert "(shard)" in r.stderr
    assert reference_expected()
    ref_path = refprobe.reference_library().path
    theirs = str(tmp_path / "client_ref")
    _build(theirs, os.path.dirname(ref_path), os.path.basename(ref_path))
    want = subprocess.run([theirs, "6200", "4200"], check=True, capture_output=True, text=True, timeout=600).stdout
    assert r.stdout == want


def test_without_peer_access_the_frame_stays_on_one_device():
    """VERDICT r4 task 2 / ADVICE: peer capability is asked when the frame is dealt out (encoder.c, hydamd_peers_reachable),
    not at the final tile.  HYDAMD_TEST_NO_P2P=1 (the HYD_TEST_HOOKS flavour of the library only) makes the probe answer "no": the frame is coded on the encoder's home device,
    says so once on stderr, and is the reference's file"""
    ours, ref, err = _run(("photo", 4296, 4168, 8), "0,0,0,0", {"HYDAMD_TEST_NO_P2P": "1", "HYDAMD_LIB": hbuild.PROBE_PATH})
    assert "(shard)" not in err, "the frame was dealt out although the devices cannot read each other"
    assert "no peer access" in err
    assert ours == ref


def test_verify_peers_checks_every_shards_view_and_names_the_pair_that_differs():
    """HYDAMD_VERIFY_PEERS=1: owner and assembling device sum every shard's view (hydamd_verify_enqueue); equal sums -> the
    reference's file as ever; a sum that differs (hook: shard 2's as read by the assembling device) fails the frame with the
    device pair in the message"""
    ours, ref, err = _run(("photo", 4296, 4168, 8), "0,0,0,0", {"HYDAMD_VERIFY_PEERS": "1"})
    assert "(shard)" in err
    assert ours == ref
    got, _, _ = _run(("photo", 4296, 4168, 8), "0,0,0,0", {"HYDAMD_VERIFY_PEERS": "1", "HYDAMD_TEST_CORRUPT_PEER_VIEW": "2", "HYDAMD_LIB": hbuild.PROBE_PATH},
                     reference=False)
    assert got.startswith("ERR") and "peer read mismatch" in got and "shard 2" in got


def test_peer_reads_verify_themselves_at_first_use_of_a_device_pair_and_are_trusted_afterwards():
    """VERDICT r5 task 3: nobody has to ask (no HYDAMD_VERIFY_PEERS): the first sharded frame over a pair of list entries
    checks its floor read and its views, later frames over the same pairs do not.  Encoders take the list's entries in turn
    as home (= assembling) device: on the list 0,0,0,0 the first three images meet new (reader, owner) pairs — views are
    read by the assembling entry, floors by every later shard from every earlier one — and after them all twelve ordered
    pairs are latched: images four and five verify nothing."""
    ours, ref, err = _run(("photo", 4296, 4168, 8), "0,0,0,0", repeat=5)
    assert ours == [ref] * 5
    per_image = err.split("== image ")[1:]
    assert len(per_image) == 5
    verified = ["peer reads verified" in t for t in per_image]
    assert verified == [True, True, True, False, False], verified
    for k, t in enumerate(per_image):
        assert f"assembled on entry {k % 4} " in t, t


def test_a_peer_read_that_fails_its_first_use_check_sends_the_frame_through_the_host_and_later_frames_to_one_device():
    """hooks of the HYD_TEST_HOOKS flavour: shard 2's view as the assembling device 'sees' it differs / the floor a shard
    'read' from its peers differs.  Default mode (first use): one line on stderr, THIS frame finished without peer reads —
    floors as host values, every shard replayed, results read from each shard's own device, assembled by the host's writers
    — and equal to the reference's; the next image of the process is not dealt out any more."""
    for hook, case in (({"HYDAMD_TEST_CORRUPT_PEER_VIEW": "2"}, ("photo", 4296, 4168, 8)),
                       ({"HYDAMD_TEST_CORRUPT_PEER_FLOOR": "1"}, ("photo", 4100, 4100, 32))):  # float samples: the floor decides log_alphabet_size
        ours, ref, err = _run(case, "0,0,0,0", dict(hook, HYDAMD_LIB=hbuild.PROBE_PATH), repeat=2)
        assert ours == [ref, ref], (hook, ours, ref)
        first, second = err.split("== image ")[1:]
        assert "(shard)" in first and "peer read mismatch" in first and "finished through host memory" in first
        assert "(shard)" not in second and "peer read mismatch" not in second


def test_two_sharded_encoders_side_by_side_assemble_on_different_entries_of_the_list():
    """concurrent sharded encoders used to pile every assembly and every read-back on the list's first device"""
    ours, ref, err = _run(("photo", 4296, 4168, 8), "0,0,0,0", threads=2)
    assert ours == [ref, ref]
    assert "assembled on entry 0 " in err and "assembled on entry 1 " in err
