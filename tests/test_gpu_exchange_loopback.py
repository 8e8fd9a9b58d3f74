"""The R > 1 branch of the exchange, EXECUTED.  RCCL refuses two ranks on one device and the pool hands out one GPU,
so until now ncclSend / ncclRecv / ncclAllGather had only ever run with world = 1.  tests/cpp/loopback_rccl.cpp is a test double for
the nine nccl* entry points the library resolves (ranks = threads of one process, a transfer = a hipMemcpyAsync on the receiver's
stream ordered by events against the sender's); ARCTIC_RCCL_LIB, honoured only when set, makes the library load it instead of RCCL.
tests/loopback_worlds.py then drives R handles the way bench.py's ranks do -- arctic_comm_init's layout all-gather, the double-
buffered arctic_gather_frame (grouped send / recv + placement kernel), the sharded shadow map's in-place all-gather -- and compares
the root's frames with the single-device frame byte for byte.  What stays unexecuted is RCCL itself (and xGMI): src/renderer/rhi.cpp:120-124
is a single adapter, the exchange is this build's generalisation."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "loopback_rccl.cpp")
LIB = os.path.join(ROOT, "tests", "cpp", "libloopback_rccl.so")


def build():
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", SRC, "-o", LIB,
                           "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-pthread"])


def test_loopback_communicator_builds_and_exports_what_the_library_resolves():
    """(CPU) the nine entry points of csrc/renderer.cpp: struct Rccl"""
    import ctypes
    build()
    L = ctypes.CDLL(LIB)
    for name in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclAllGather", "ncclSend", "ncclRecv", "ncclGroupStart", "ncclGroupEnd", "ncclGetErrorString"):
        assert hasattr(L, name), name


@pytest.mark.gpu
@pytest.mark.parametrize("world,layout,shadow,streams", [(2, "bands", False, False), (3, "bands", True, False), (8, "bands", False, False), (2, "rows", True, False),
                                                         (3, "rows", False, False), (8, "rows", True, False),
                                                         # every rank on a caller's stream, as bench.py's ranks are: own stream = exchange stream only
                                                         (3, "bands", False, True), (4, "rows", True, True),
                                                         # ... and back to the handle's own stream between frames, a gather in flight on the stream the exchange borrowed
                                                         (3, "bands", False, "ownback"), (2, "rows", True, "ownback")],
                         ids=lambda v: str(v))
def test_exchange_with_more_than_one_rank(hip, world, layout, shadow, streams):
    build()
    env = dict(os.environ, ARCTIC_RCCL_LIB=LIB)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "loopback_worlds.py"), str(world), layout] + (["shadow"] if shadow else []) + (["ownback"] if streams == "ownback" else ["streams"] if streams else []),
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "LOOPBACK_OK" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]


@pytest.mark.gpu
def test_override_is_honoured_only_when_set_and_never_falls_back(hip):
    """ARCTIC_RCCL_LIB naming a file that does not exist: comm calls fail loudly instead of quietly loading RCCL"""
    code = ("import sys; sys.path.insert(0, %r); import __graft_entry__ as e; pkg = e.load_package()\n"
            "try:\n    pkg.renderer.Renderer.comm_unique_id(); print('LOADED')\nexcept Exception as exc:\n    print('REFUSED', exc)\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, ARCTIC_RCCL_LIB="/nonexistent/librccl.so"))
    assert "REFUSED" in out.stdout, out.stdout + out.stderr
