"""Which engine moves a large device-to-host copy on this box?  Run with AMD_LOG_LEVEL=4 and grep the runtime's log
for 'HSA Copy' / 'Blit'; prints the copy's rate with and without kernels running beside it."""
import sys
import time

import torch

n = 51 << 20
src = torch.empty(n, dtype=torch.uint8, device="cuda")
dst = torch.empty(n, dtype=torch.uint8).pin_memory()
st = torch.cuda.Stream()
torch.cuda.synchronize()
for busy in (False, True):
    a = torch.randn(8192, 8192, device="cuda")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if busy:
        for _ in range(20):
            a = a * 1.0001 + 0.5  # elementwise kernels on the default stream
    with torch.cuda.stream(st):
        for _ in range(5):
            dst.copy_(src, non_blocking=True)
    st.synchronize()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"busy={busy}: 5 x {n >> 20} MB D2H in {dt * 1e3:.2f} ms = {5 * n / dt / 1e9:.1f} GB/s", flush=True)
