"""Tile-mode frames through the drop-in API (tile_size_shift >= 0: one frame per tile, each hyd_send_tile call ends with its
frame's bytes, reference encoder.c:339-378): time per image and per tile for the tile sizes the format allows."""
import ctypes
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from hydrium_amd import api, synth

size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 8
img = np.ascontiguousarray(synth.make_image("photo", size, size, depth))
lib = api.Library()
big = (ctypes.c_uint8 * (64 << 20))()
for shift in (-1, 3, 2, 1, 0):
    kw = {} if shift < 0 else dict(shift_x=shift, shift_y=shift)
    times = []
    for i in range(4):
        t = time.perf_counter()
        d = api.encode_image(lib, img, out_buf=big, **kw)
        times.append((time.perf_counter() - t) * 1e3)
    tw, th = api.tile_dims(size, size, shift, shift)
    ntiles = (-(-size // tw)) * (-(-size // th))
    best = min(times[1:])
    print(f"shift {shift:2d}: tiles of {tw}x{th}, {ntiles:4d} calls, {best:8.2f} ms per image, {best / ntiles:6.3f} ms per tile, "
          f"{size * size / best / 1e3:7.1f} Mpixel/s, {len(d)} bytes {hashlib.md5(d).hexdigest()[:8]}", flush=True)
