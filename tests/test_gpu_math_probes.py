"""The square roots of the numeric-Jacobian paths are claimed to be the SAME correctly rounded numbers as sqrt() (device_math.h:
sqrt_ieee_unscaled, sqrt_ieee_near).  tools/sqrt_probe.hip checks that on the device over 4e8 arguments; this test runs it."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sqrt_variants_are_bit_identical_to_sqrt(gpu):
    exe = os.path.join(ROOT, "tools", "sqrt_probe.bin")
    if not os.path.exists(exe):   # (__graft_entry__.build() compiles it; the binary travels with the snapshot)
        hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-I", os.path.join(ROOT, "localization_amd", "csrc"),
                               os.path.join(ROOT, "tools", "sqrt_probe.hip"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    lines = [l for l in out.stdout.splitlines() if "mismatches" in l]
    assert len(lines) == 4, out.stdout
    for l in lines:
        assert ": 0 mismatches" in l, l
    assert "DIFFERENT" not in out.stdout and out.stdout.count("same bits") == 8


def test_single_axis_oplus_is_the_textbook_evaluation(gpu):
    """numeric_jacobian.h: oplus_axis_plain (X * fromVectorMQT(+-1e-9 e_d) without the arithmetic on the increment's exact zeros and
    ones) against the textbook evaluation, bit for bit up to the sign of exact zeros — tools/oplus_probe.hip on the device."""
    exe = os.path.join(ROOT, "tools", "oplus_probe.bin")
    if not os.path.exists(exe):
        hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-I", os.path.join(ROOT, "localization_amd", "csrc"),
                               os.path.join(ROOT, "tools", "oplus_probe.hip"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and ": 0 mismatches" in out.stdout, (out.returncode, out.stdout, out.stderr)
