"""How fast can 8K frames' worth of pinned host memory reach the GPU?  One stream of 24 MB copies (what the drop-in API
issues per tile), two streams side by side, one large copy, and a kernel that reads the pinned memory itself.
Run a second time with HSA_ENABLE_SDMA=0 for shader copies instead of the DMA engines."""
import os
import time

import torch

tile = 2048 * 2048 * 3 * 2
tiles = 16
host = [torch.empty(tile, dtype=torch.uint8).pin_memory() for _ in range(tiles)]
dev = [torch.empty(tile, dtype=torch.uint8, device="cuda") for _ in range(tiles)]
big_h = torch.empty(tile * tiles, dtype=torch.uint8).pin_memory()
big_d = torch.empty(tile * tiles, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()


def timed(label, fn, nbytes, reps=5):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(f"{label:44s} {best * 1e3:7.2f} ms  {nbytes / best / 1e9:6.1f} GB/s", flush=True)


def one_stream():
    with torch.cuda.stream(s1):
        for h, d in zip(host, dev):
            d.copy_(h, non_blocking=True)


def two_streams():
    for i, (h, d) in enumerate(zip(host, dev)):
        with torch.cuda.stream(s1 if i & 1 else s2):
            d.copy_(h, non_blocking=True)


def one_big():
    with torch.cuda.stream(s1):
        big_d.copy_(big_h, non_blocking=True)


def halves():
    n = tile * tiles // 2
    with torch.cuda.stream(s1):
        big_d[:n].copy_(big_h[:n], non_blocking=True)
    with torch.cuda.stream(s2):
        big_d[n:].copy_(big_h[n:], non_blocking=True)


print("HSA_ENABLE_SDMA =", os.environ.get("HSA_ENABLE_SDMA", "(default)"))
timed("16 x 24 MB, one stream", one_stream, tile * tiles)
timed("16 x 24 MB, alternating between two streams", two_streams, tile * tiles)
timed("one 403 MB copy", one_big, tile * tiles)
timed("two 201 MB copies on two streams", halves, tile * tiles)
