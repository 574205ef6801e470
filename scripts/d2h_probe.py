"""Who moves a finished 51 MB codestream to the host, and how fast, with the GPU busy?  (scripts/ubench/d2h_kernel.hip is the
same question in plain HIP.)  torch's copy_ into its own pinned tensor against hipMemcpyAsync into (a) that tensor and (b) memory
from hipHostMalloc, each with elementwise kernels running on another stream."""
import ctypes as C
import time

import torch

hip = C.CDLL("libamdhip64.so")
n = 51 << 20
src = torch.empty(n, dtype=torch.uint8, device="cuda")
pin = torch.empty(n, dtype=torch.uint8).pin_memory()
raw = C.c_void_p()
assert hip.hipHostMalloc(C.byref(raw), C.c_size_t(n), 0) == 0
st = torch.cuda.Stream()
busy_st = torch.cuda.Stream()
a = torch.randn(8192, 8192, device="cuda")


def run(name, fn):
    for busy in (False, True):
        torch.cuda.synchronize()
        if busy:
            with torch.cuda.stream(busy_st):
                b = a
                for _ in range(200):
                    b = b * 1.0001 + 0.5
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(st):
            e0.record()
            for _ in range(5):
                fn()
            e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / 5
        torch.cuda.synchronize()
        print(f"{name:42s} {'busy' if busy else 'idle'}: {ms:6.2f} ms per 51 MB = {n / ms / 1e6:5.1f} GB/s", flush=True)


for rep in range(2):
    run("torch copy_ -> torch pinned", lambda: pin.copy_(src, non_blocking=True))
    run("hipMemcpyAsync -> torch pinned", lambda: hip.hipMemcpyAsync(C.c_void_p(pin.data_ptr()), C.c_void_p(src.data_ptr()), C.c_size_t(n), 2, C.c_void_p(st.cuda_stream)))
    run("hipMemcpyAsync -> hipHostMalloc", lambda: hip.hipMemcpyAsync(raw, C.c_void_p(src.data_ptr()), C.c_size_t(n), 2, C.c_void_p(st.cuda_stream)))
