python -m pytest tests/test_gpu_api_parity.py tests/test_gpu_assembler.py tests/test_gpu_multi_device.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -3
taskset -c 0-63,128-191 python scripts/api_variance.py 5 2>&1 | tail -3 | cut -c1-120
python - <<'PY'
import ctypes, time, hashlib, sys
sys.path.insert(0,'.')
import numpy as np
from hydrium_amd import api, synth, placement
placement.bind_near_gpu(0)
img = np.ascontiguousarray(synth.make_image("photo", 8192, 8192, 16, device="cuda").cpu().numpy().view(np.uint16))
lib = api.Library(); big = (ctypes.c_uint8 * (32 << 20))()
api.encode_image(lib, img, out_buf=big)
ts=[]
for _ in range(7):
    t=time.perf_counter(); d=api.encode_image(lib, img, out_buf=big, in_place=True); ts.append(time.perf_counter()-t)
print("api 32 MiB buffer, ms:", [round(x*1e3,2) for x in ts], hashlib.md5(bytes(d)).hexdigest())
PY
