"""The C-ABI shared library on a machine without a GPU: it loads, exports every symbol include/arctic_hip.h
declares, its host-only entry point agrees with the oracle bit for bit, and it refuses to run without a device
(there is no CPU fallback to fall into)."""
import ctypes as C
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def lib(pkg):
    from importlib import import_module
    b = import_module("arctic_renderer_amd.binding")
    if not os.path.exists(b.LIB_PATH):
        import __graft_entry__ as entry
        entry.build()
    return b


def test_library_exports_every_declared_symbol(lib):
    L = lib.lib()
    names = lib.header_symbols()
    assert len(names) >= 25 and "arctic_render_frame" in names and "arctic_create_material" in names
    for n in names:
        assert hasattr(L, n), n
    assert set(lib.SIGNATURES) == set(names), "binding.py and arctic_hip.h disagree"
    assert L.arctic_version() >= 100


def test_pod_layouts_match_the_reference_structs(pkg):
    S = pkg.scene
    assert S.VERTEX_DTYPE.itemsize == 56        # scene.hpp:40-47, 14 floats
    assert S.LIGHT_DTYPE.itemsize == 32         # scene.hpp:88-94
    assert S.OBJECT_DTYPE.itemsize == 72        # mat4 + size_t
    assert C.sizeof(S.CCamera) == 36 and C.sizeof(S.CDirectionalLight) == 32 and C.sizeof(S.CSettings) == 12
    assert S.CScene.point_lights.offset == 72 and S.CScene.objects.offset == 88 and C.sizeof(S.CScene) == 104


def test_frame_constants_bit_exact_vs_oracle(lib, pkg, oracle):
    """arctic_frame_constants is host code (glm-equivalent builders): must equal the oracle's to the bit, because
    the visibility pass is bit-exact only if both start from the same matrices."""
    from importlib import import_module
    R = import_module("arctic_renderer_amd.renderer")
    rng = np.random.default_rng(11)
    for _ in range(50):
        desc = pkg.scene.SceneDesc(
            camera=dict(eye=rng.uniform(-20, 20, 3), rotation=(rng.uniform(-89, 89), rng.uniform(-180, 180)), aspect=rng.uniform(0.5, 2.5),
                        fov_y=rng.uniform(20, 100), z_near_far=(rng.uniform(0.01, 1.0), rng.uniform(50, 2000))),
            ambient=0.1, sun=dict(position=rng.uniform(-40, 40, 3), rotation=(rng.uniform(-89, -10), rng.uniform(-180, 180)), color=(8, 8, 8)),
            objects=np.zeros(0, pkg.scene.OBJECT_DTYPE))
        a = R.frame_constants(desc)
        b = oracle.frame_constants(desc)
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x.view(np.uint32), y.view(np.uint32))


def test_create_fails_loudly_without_a_device(lib, pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    from importlib import import_module
    R = import_module("arctic_renderer_amd.renderer")
    with pytest.raises(R.ArcticError) as e:
        R.Renderer(64, 64, 0, 16)
    assert "no HIP device" in str(e.value)


def test_bad_arguments_are_rejected_before_touching_the_device(lib):
    L = lib.lib()
    from arctic_renderer_amd.scene import CCreateInfo
    err = C.create_string_buffer(256)
    assert not L.arctic_create(None, err, 256)
    info = CCreateInfo(0, 64, 0, 16, 0, 0, 0)
    assert not L.arctic_create(C.byref(info), err, 256) and b"width/height" in err.value
    info = CCreateInfo(64, 64, 0, 16, 0, 40, 20)
    assert not L.arctic_create(C.byref(info), err, 256) and b"row shard" in err.value
    assert L.arctic_flush(None) == -1 and L.arctic_stats(None, None, 0) == -1
    assert L.arctic_last_error(None) == b"null handle"
