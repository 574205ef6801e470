python scripts/k1_variants.py --run --rounds 1 nog 2>&1 | tail -3
for lib in before_u8 nog; do
HYDAMD_LIB=$PWD/scripts/probe_build/k1v_$lib.so python - <<PY
import torch, hashlib
from hydrium_amd import device, synth
img = synth.make_image("photo", 8192, 8192, 8, device=torch.device("cuda", 0))
with device.DeviceContext(0, 16, 0) as ctx:
    ctx.set_rans_waves(5); ctx.set_lf_coder(0)
    ctx.encode_image_tensor(img); ctx.sync()
    md5 = hashlib.md5(ctx.read_payload()).hexdigest()[:12]
    ctx.profile(True)
    for _ in range(6):
        ctx.encode_image_tensor(img); ctx.sync()
    ms, n = ctx.profile_read()["transform_tokenize"]
    print("$lib RGB8 8192x8192: K1", round(ms / n, 4), "ms, sections md5", md5)
PY
done
timeout 900 python -m pytest tests/test_gpu_device_parity.py tests/test_gpu_api_parity.py -x -q 2>&1 | grep -E "passed|failed"
