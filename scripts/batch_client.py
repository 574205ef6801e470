"""Build and run scripts/batch_client.c (the 4K batch of BASELINE configs[4] through the drop-in API from C threads) on the
frame bench.py's batch legs use.  usage: python scripts/batch_client.py [threads ...]     (default 8 10 12 16 24)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "22")
from hydrium_amd import build as hb, placement, synth  # noqa: E402

hb.build()
exe = os.path.join(ROOT, "scripts", "probe_build", "batch_client")
os.makedirs(os.path.dirname(exe), exist_ok=True)
libdir = os.path.dirname(hb.LIB_PATH)
subprocess.run(["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", f"-I{os.path.join(ROOT, 'include')}", os.path.join(ROOT, "scripts", "batch_client.c"),
                "-o", exe, f"-L{libdir}", f"-l:{os.path.basename(hb.LIB_PATH)}", f"-Wl,-rpath,{libdir}", "-lpthread"], check=True)
raw = "/tmp/batch_frame.rgb"
synth.make_image("photo", 3840, 2160, 8, seed=1234).tofile(raw)
import torch  # noqa: E402  (only to find the GPU's NUMA node)

print("bound to", placement.bind_near_gpu(0), flush=True)
for t in [int(x) for x in sys.argv[1:]] or [8, 10, 12, 16, 24]:
    r = subprocess.run([exe, raw, "3840", "2160", str(t), "512"], capture_output=True, text=True)
    print(r.stdout.strip().splitlines()[-1] if r.returncode == 0 else f"threads {t}: FAILED {r.stderr[-300:]}", flush=True)
