"""A 1024x1024 RGB8 image in 256x256 tiles through the drop-in API, three times: the command rocprofv3 wraps to see what one
tile-mode frame's 1.66 ms are made of (scripts/rocpd_timeline.py on the result)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hydrium_amd import api, synth
img = np.ascontiguousarray(synth.make_image("photo", 1024, 1024, 8))
lib = api.Library()
big = (ctypes.c_uint8 * (8 << 20))()
for _ in range(3):
    api.encode_image(lib, img, out_buf=big, shift_x=0, shift_y=0)
