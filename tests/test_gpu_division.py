"""The short exact division of shadow_coords.h (quotient_near_one: x / w for a divisor within six ulp-steps of 1.0, forward.hlsl:70 under the
reference's orthographic sun, scene.cpp:61-70) against the compiler's IEEE division -- EXHAUSTIVELY: all 2^32 numerators for each of the 13
divisors, on the device (needs an MI355X; about a second).  The quotient feeds floor() and the 25 PCF compares, so "close" is not enough."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_quotient_near_one_is_the_ieee_quotient_for_every_numerator(hip, tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "div_near_one")
    subprocess.check_call([hipcc, "-O3", "-ffp-contract=off", "-Wno-unused-value", "--offload-arch=gfx950",
                           os.path.join(ROOT, "tools", "experiments", "div_near_one.hip"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    lines = [l for l in out.stdout.splitlines() if l.startswith("w = ")]
    assert out.returncode == 0 and len(lines) == 13, out.stdout[-3000:] + out.stderr[-2000:]
    for l in lines:     # "A0 (raw rcp, 1 correction): 0 (+N out of range)": what the kernels run, inside the kernels' own guard
        assert "A0 (raw rcp, 1 correction): 0 (" in l, l
