"""The C++ host-side mirror (arctic-renderer_amd/host/renderer.hpp): compiles against the C-ABI with plain g++ on any
machine, and on a GPU box renders a frame driven exactly like src/app.cpp drives the reference Renderer."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "arctic-renderer_amd", "csrc")
EXE = os.path.join(ROOT, "tests", "cpp", "mirror_smoke")


def build():
    subprocess.check_call(["g++", "-std=c++20", "-O1", "-Wall", "-Wextra", os.path.join(ROOT, "tests", "cpp", "mirror_smoke.cpp"), "-o", EXE,
                           "-L" + CSRC, "-larctic_hip", "-Wl,-rpath," + CSRC, "-Wl,-rpath,/opt/rocm/lib"])


def test_mirror_compiles_and_links(pkg):
    build()
    import torch
    if not torch.cuda.is_available():
        # no device: init() must fail loudly (exit 0 only because we pass the "expect failure" flag)
        out = subprocess.run([EXE, "expect-no-device"], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0 and "no HIP device" in out.stdout


@pytest.mark.gpu
def test_mirror_renders_a_frame(pkg, hip):
    build()
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "checksum" in out.stdout
