"""CPU-only: what the compiled kernels ask of a compute unit, read from the code object hipcc cross-compiles here.

Round 6's find: a kernel whose STATIC LDS lets only two of its workgroups onto a compute unit is given a padded
register allocation by the compiler (.amdhsa_next_free_vgpr 257 for the lane-form chain kernel, which uses 164) - harmless for
the kernel itself, half a SIMD lost to every other kernel's wavefronts.  Five rounds of placement experiments assumed the
allocation was the use.  This test pins the numbers the design's co-residency arithmetic rests on, so that a toolchain or source
change that moves them is seen here and not three probes later."""
import os
import re
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _descriptors(flags=()):
    from hydrium_amd import build as hb

    if not os.path.exists(hb.HIPCC):
        pytest.skip("no hipcc")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.run([hb.HIPCC] + hb.HIP_FLAGS + list(flags) + ["--cuda-device-only", "-S", "-o", out, os.path.join(hb.CSRC, "hip", "kernels.hip")],
                       check=True, capture_output=True)
        text = open(out).read()
    res = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S):
        body = m.group(2)
        res[m.group(1)] = {k: int(re.search(r"\.amdhsa_" + k + r" (\d+)", body).group(1))
                           for k in ("group_segment_fixed_size", "next_free_vgpr", "private_segment_fixed_size")}
    return res


def _one(res, prefix):
    hits = [v for k, v in res.items() if k.startswith(prefix)]
    assert len(hits) == 1, (prefix, [k for k in res if k.startswith(prefix)])
    return hits[0]


def test_what_the_shipped_kernels_ask_of_a_compute_unit():
    res = _descriptors()
    k1 = _one(res, "_Z20k_transform_tokenizeILi1ELi0E")  # 16-bit samples, curves in registers + one gather: the contract workload's kernel
    # four workgroups per compute unit: 4 x 25 granules of 1280 B of LDS, 4 wavefronts x 120 registers per SIMD
    assert k1["group_segment_fixed_size"] <= 25 * 1280 and k1["next_free_vgpr"] <= 120 and k1["private_segment_fixed_size"] == 0, k1
    chain = _one(res, "_Z12k_rans_lanesILi9E")
    assert chain["group_segment_fixed_size"] <= 63 * 1280 and chain["private_segment_fixed_size"] == 0, chain
    # the compiler's padding: static LDS bounds the kernel to one wavefront per SIMD, so it is GIVEN the registers that rule out a
    # second (512 / 2 + 1); what the kernel uses is 164.  If this ever reads 164, the chain has stopped taking half a SIMD from
    # the transform wavefronts beside it: re-measure the loop
    assert chain["next_free_vgpr"] == 257, chain
    for nc in (3, 2, 1):  # smaller tables (16384^2 frames have three clusters per preset): more workgroups fit, less or no padding
        assert _one(res, f"_Z12k_rans_lanesILi{nc}E")["next_free_vgpr"] <= 169
