"""The float tier's C-library calls, pinned: libxaac_amd/csrc/fx_libm.h holds the four expressions in which Path A (eSBR
pre-flattening, the PVC envelope decoder) goes through double log10 / pow and rounds to float.  The reference gets glibc's,
the kernels get ROCm's device library's; neither is correctly rounded.  tests/fuzz/libm_probe.hip evaluates each expression
on the GPU and on the host (the same inline functions, compiled for both) over EVERY float input its call site can produce
-- 1.1e9 / 2.3e9 / 1.1e9 / 2.2e9 inputs -- and compares the float words.  One differing word fails the test: the claim
"Path A is word-identical to the reference" then no longer rests on "none met so far".
Measured on MI355X / ROCm 7.2 / glibc 2.35: 0 differing words in all four sweeps (20 s)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def probe(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("libm") / "libm_probe")
    # the product's flags (libxaac_amd/csrc/Makefile): -O3 -std=c++17 -ffp-contract=off, gfx950
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-unused-value",
                           os.path.join(ROOT, "tests", "fuzz", "libm_probe.hip"), "-o", exe, "-lpthread"])
    return exe


@pytest.mark.gpu
@pytest.mark.parametrize("fn, name, at_least", [(0, "xm_log10f_of", 1_100_000_000), (1, "xm_pow10_tenth", 2_290_000_000),
                                                (2, "xm_10log10f_of", 1_070_000_000), (3, "xm_pow10f_of", 2_240_000_000)])
def test_device_libm_equals_host_libm_on_every_reachable_float(probe, fn, name, at_least):
    p = subprocess.run([probe, str(fn), "16"], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    m = re.search(r"^%s inputs (\d+) differing (\d+)$" % name, p.stdout, re.M)
    assert m, p.stdout[-500:]
    assert int(m.group(1)) >= at_least
    assert int(m.group(2)) == 0, "device and host libm disagree:\n" + p.stdout[-1500:]
