"""GPU: a seeded sweep of whole files through the drop-in API against the compiled reference (scripts/fuzz_api_parity.py):
ragged sizes from 1 to 2400 pixels, 8/16-bit and float input, dark 16-bit content that mixes both branches of the transfer
curve inside a wavefront, linear light, every tile shift, planar and bottom-up layouts — under both forms of the entropy
stage — and a handful of multi-LF-group frames (which the device-side assembler puts together).  A failure prints the
seed and case number to replay with the script."""
import importlib.util
import os

import pytest

from conftest import has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]


def _fuzz():
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "fuzz_api_parity.py")
    spec = importlib.util.spec_from_file_location("fuzz_api_parity", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("form,seed", [(4, 301), (5, 302)])
def test_seeded_sweep_against_the_reference(ref_lib, monkeypatch, form, seed):
    from hydrium_amd import api

    lib = api.Library()
    monkeypatch.setenv("HYDAMD_RANS_WAVES", str(form))
    lib.dll.hydamd_trim_cache()  # parked contexts keep the form they were created with
    try:
        ran, bad = _fuzz().sweep(140, seed, lib=lib, ref=ref_lib, budget_s=22)
        big_ran, big_bad = _fuzz().sweep(6, seed + 50, large=True, lib=lib, ref=ref_lib, budget_s=12)
    finally:
        lib.dll.hydamd_trim_cache()
    assert not bad and not big_bad, "\n".join(bad + big_bad)
    assert ran >= 60 and big_ran >= 2, (ran, big_ran)  # the time budget must not hollow the sweep out
