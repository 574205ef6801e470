import functools
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: multi-second CPU test")


@functools.lru_cache(maxsize=32)
def _image(kind, w, h, depth, seed):
    from hydrium_amd import synth

    img = synth.make_image(kind, w, h, depth, seed)
    img.setflags(write=False)
    return img


@pytest.fixture(scope="session")
def image():
    """image(kind, w, h, depth=8, seed=1234) -> cached read-only (h, w, 3) array."""

    def get(kind, w, h, depth=8, seed=1234):
        return _image(kind, w, h, depth, seed)

    return get


@pytest.fixture(scope="session")
def ref_probe():
    from oracle import refprobe

    refprobe.build()
    if not refprobe.available():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    return refprobe.Probe()


@pytest.fixture(scope="session")
def ref_lib():
    from oracle import refprobe

    refprobe.build()
    if not refprobe.available():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    return refprobe.reference_library()


def has_gpu() -> bool:
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False
