"""The synthetic scene generator (stand-in for App::load_scene): deterministic, well formed, sized as SURVEY 8(d) says."""
import hashlib

import numpy as np


def digest(sc):
    h = hashlib.sha256()
    for d, n, m in sc.materials:
        h.update(d.tobytes()); h.update(n.tobytes()); h.update(m.tobytes())
    for v, i, mat in sc.meshes:
        h.update(v.tobytes()); h.update(i.tobytes()); h.update(bytes([mat]))
    h.update(sc.lights.tobytes()); h.update(sc.desc.objects.tobytes())
    return h.hexdigest()


def test_generator_is_deterministic(pkg):
    a, b = pkg.scenes.config3(scale=0.1, tex=64), pkg.scenes.config3(scale=0.1, tex=64)
    assert digest(a) == digest(b)
    assert digest(a) != digest(pkg.scenes.config4(scale=0.1, tex=64))


def test_config_shapes(pkg):
    c1, c2, c3 = (pkg.scenes.CONFIGS[k](scale=0.1, tex=32) for k in (1, 2, 3))
    assert (c1.shadow_size, len(c1.lights), len(c1.materials), c1.settings[0]) == (0, 0, 1, 0)
    assert (len(c2.materials), len(c2.meshes), len(c2.lights), c2.settings[0]) == (6, 6, 0, 0) and c2.shadow_size > 0
    assert (len(c3.materials), len(c3.lights), c3.settings[0]) == (25, 64, 2)
    assert len(pkg.scenes.config4(scale=0.05, tex=32).lights) == 256
    for sc in (c1, c2, c3):
        assert sc.width % 8 == 0 and sc.height % 8 == 0
        for v, i, mat in sc.meshes:
            assert i.max() < len(v) and len(i) % 3 == 0 and mat < len(sc.materials)
            for k in ("normal", "tangent", "bitangent"):
                np.testing.assert_allclose(np.linalg.norm(v[k], axis=1), 1.0, atol=1e-5)


def test_texture_statistics_follow_the_survey(pkg):
    rng = np.random.default_rng(5)
    d, n, m = pkg.scenes.make_material_textures(rng, 128)
    assert d[..., :3].min() >= 30 and d[..., :3].max() <= 230 and (d[..., 3] == 255).all()
    assert abs(int(n[..., 0].min()) - 128) <= 41 and abs(int(n[..., 0].max()) - 128) <= 41 and (n[..., 2] == 255).all()
    assert m[..., 1].min() >= 13 and set(np.unique(m[..., 2])) <= {0, 255}
    assert 0.1 < (m[..., 2] == 255).mean() < 0.3


def test_full_size_dimensions(pkg):
    # only the cheap header of the full-size configs (no texture generation): sizes the metric is quoted on
    assert pkg.scenes._dims(3840, 2160, 1.0) == (3840, 2160) and pkg.scenes._dims(7680, 4320, 1.0) == (7680, 4320)
