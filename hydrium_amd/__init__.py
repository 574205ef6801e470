"""hydrium_amd — MI355X-native build of hydrium's JPEG XL tile encoder.

The product is the C-ABI shared library ``hydrium_amd/lib/libhydrium.so.0`` (drop-in for the
reference's libhydrium, plus the additive ``hydamd_*`` device-level entry points).  The Python
modules here are thin plumbing over it for tests, ``bench.py`` and multi-GPU drivers:

* ``api``      — ctypes binding of the nine ``hyd_*`` functions
* ``device``   — ctypes binding of the ``hydamd_*`` functions (device pointers from torch tensors)
* ``sharding`` — LF-group partitioning and the RCCL all-gather of coded sections
* ``synth``    — deterministic synthetic images
* ``build``    — in-tree build (hipcc + gcc)
"""


def preload_hip_runtime() -> None:
    """Make the process use ONE HIP runtime.

    PyTorch-ROCm bundles its own libamdhip64.so.7; libhydrium.so.0 links the system one with the
    same soname.  Whichever is loaded first serves both, and a process that loads the system
    runtime first and torch's HSA stack second ends up with no visible device.  Importing torch
    before dlopen()ing our library keeps everything on torch's runtime.  A pure C caller is not
    affected (it only ever has the system runtime).
    """
    try:
        import torch  # noqa: F401
    except Exception:  # torch absent: nothing to keep consistent
        pass
