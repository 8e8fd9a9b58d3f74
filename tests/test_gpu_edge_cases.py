"""Edge cases through the C-ABI (GPU): empty and degenerate inputs, extreme sizes, ragged frames -- each still compared
with the oracle under the same bars as tests/test_gpu_parity.py."""
import copy

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-4


def tri_mesh(pkg, pts, uv=None, normal=(0, 0, 1)):
    v = np.zeros(len(pts), pkg.scene.VERTEX_DTYPE)
    v["position"] = pts
    v["normal"], v["tangent"], v["bitangent"] = normal, (1, 0, 0), (0, 1, 0)
    v["tex_coords"] = uv if uv is not None else np.asarray(pts, np.float32)[:, :2]
    return v


def camera_scene(pkg, objects, w, h, lights=None):
    S = pkg.scene
    return S.SceneDesc(camera=dict(eye=(0, 0, 4.0), rotation=(0.0, -90.0), aspect=w / h, fov_y=60.0, z_near_far=(0.1, 100.0)),
                       ambient=0.1, sun=dict(position=(2, 10, 6), rotation=(-55.0, -110.0), color=(8, 8, 8)), objects=objects,
                       point_lights=lights)


def both(pkg, oracle, hip, w, h, shadow, mats, meshes, objs, lights, settings=(0, 2.2, 1.0)):
    outs = []
    for cls in (oracle.Oracle, hip.Renderer):
        r = cls(w, h, shadow, 16)
        for m in mats:
            r.create_material(*m)
        for v, i, mat in meshes:
            r.create_mesh(v, i, mat)
        r.update_lights(lights)
        if cls is hip.Renderer:
            r.set_option("keep_float_output", 1)
        desc = camera_scene(pkg, pkg.scene.make_objects(objs), w, h, lights)
        img = r.render_frame(desc, settings)
        outs.append((img, r.read_output()[0], r.read_gbuffer(), r))
    return outs


def check(outs):
    (oi, ol, og, o), (hi, hl, hg, r) = outs
    for a, b in zip(og, hg):
        np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))
    assert np.abs(ol - hl).max() <= TOL
    assert np.abs(oi.astype(np.int16) - hi.astype(np.int16)).max() <= 1
    r.close()
    o.close()


def test_empty_scene_is_black(pkg, oracle, hip):
    outs = both(pkg, oracle, hip, 40, 24, 64, [pkg.scenes.fallback_textures()], [], [], np.zeros(0, pkg.scene.LIGHT_DTYPE))
    assert (outs[1][0][..., :3] == 0).all() and (outs[1][0][..., 3] == 255).all()
    check(outs)


def test_ragged_frame_sizes(pkg, oracle, hip):
    """width / height that are not multiples of the 8-pixel tile, down to a single pixel."""
    rng = np.random.default_rng(8)
    mats = [pkg.scenes.make_material_textures(rng, 32)]
    quad = tri_mesh(pkg, [(-3, -2, 0), (3, -2, 0), (3, 2, 0), (-3, 2, 0)])
    lights = pkg.scenes.random_lights(rng, 3, (-2, -2, 0.5), (2, 2, 3))
    for w, h in ((1, 1), (7, 5), (9, 17), (33, 8), (130, 71)):
        check(both(pkg, oracle, hip, w, h, 96, mats, [(quad, [0, 1, 2, 0, 2, 3], 0)], [(np.eye(4), 0)], lights))


def test_degenerate_and_huge_triangles(pkg, oracle, hip):
    rng = np.random.default_rng(9)
    mats = [pkg.scenes.make_material_textures(rng, 16)]
    pts = [(-1, -1, 0), (1, -1, 0), (0, 1, 0),                 # ordinary
           (0.5, 0.5, 1), (0.5, 0.5, 1), (0.5, 0.5, 1),        # zero area: three equal vertices
           (-2, 0, 1), (0, 0, 1), (2, 0, 1),                   # zero area: collinear
           (-4000, -3000, -50), (4000, -3000, -50), (0, 5000, -50),   # far bigger than the guard band
           (-0.001, -0.001, 2), (0.001, -0.001, 2), (0, 0.001, 2)]    # smaller than a pixel
    idx = list(range(15))
    lights = pkg.scenes.random_lights(rng, 2, (-2, -2, 0.5), (2, 2, 3))
    check(both(pkg, oracle, hip, 96, 64, 128, mats, [(tri_mesh(pkg, pts), idx, 0)], [(np.eye(4), 0)], lights, settings=(2, 2.2, 1.0)))


def test_triangles_beyond_the_binary64_range_take_the_integer_rasteriser(pkg, oracle, hip):
    """the rasteriser evaluates edge functions as binary64 planes when every snapped coordinate is below 2^24 (exact), in
    64-bit integers otherwise: a 3840-pixel-wide target with triangles clipped at the +-64 w guard band reaches
    64 * 1920 * 256 = 2^24.9, next to ordinary triangles in the same frame -- both paths in one item stream."""
    rng = np.random.default_rng(19)
    mats = [pkg.scenes.make_material_textures(rng, 16)]
    pts = [(-50000, -40000, -6), (50000, -40000, -6), (0, 60000, -6),     # NDC +-140: cut at the guard band on both sides
           (-30000, -2, -3), (30000, -2, -3), (0, 1.5, -3),                # a long thin one in front of it
           (-6, -2, 0), (6, -2, 0), (0, 2, 0),                             # ordinary
           (100, -1, 1), (140, -1, 1), (120, 1, 1)]                        # ordinary, near the right edge of the wide frame
    idx = list(range(12))
    lights = pkg.scenes.random_lights(rng, 2, (-2, -2, 0.5), (2, 2, 3))
    outs = both(pkg, oracle, hip, 3840, 64, 128, mats, [(tri_mesh(pkg, pts), idx, 0)], [(np.eye(4), 0)], lights)
    assert (outs[0][2][1] != 0xFFFFFFFF).mean() > 0.5
    check(outs)


def test_one_texel_textures_and_wrapping_uv(pkg, oracle, hip):
    """1x1 and 2x3 textures, texture coordinates far outside [0,1] and negative (WRAP)."""
    rng = np.random.default_rng(10)
    one = tuple(np.array([[c]], np.uint8) for c in ((200, 100, 50, 255), (140, 120, 250, 255), (255, 180, 255, 255)))
    small = tuple(rng.integers(0, 256, (3, 2, 4), dtype=np.uint8) for _ in range(3))
    uv = np.array([(-37.25, 12.5), (41.0, 12.5), (41.0, -19.75), (-37.25, -19.75)], np.float32)
    quad = tri_mesh(pkg, [(-3, -2, 0), (3, -2, 0), (3, 2, 0), (-3, 2, 0)], uv=uv)
    lights = pkg.scenes.random_lights(rng, 1, (-1, -1, 1), (1, 1, 2))
    for mat in (one, small):
        check(both(pkg, oracle, hip, 64, 48, 0, [mat], [(quad, [0, 1, 2, 0, 2, 3], 0)], [(np.eye(4), 0)], lights))


def test_sixteen_lights_and_light_cap(pkg, oracle, hip):
    """the reference's own maximum (16, renderer.hpp:22): more lights than the cap are dropped the same way on both sides."""
    rng = np.random.default_rng(12)
    mats = [pkg.scenes.make_material_textures(rng, 32)]
    quad = tri_mesh(pkg, [(-3, -2, 0), (3, -2, 0), (3, 2, 0), (-3, 2, 0)])
    lights = pkg.scenes.random_lights(rng, 23, (-3, -2, 0.2), (3, 2, 3))       # 23 > 16: clamped (renderer.cpp:587-588)
    check(both(pkg, oracle, hip, 80, 56, 128, mats, [(quad, [0, 1, 2, 0, 2, 3], 0)], [(np.eye(4), 0)], lights, settings=(1, 1.8, 0.6)))


def test_object_transforms_and_shared_meshes(pkg, oracle, hip):
    """several objects instancing one mesh with different TRS matrices (Object{trs, mesh_idx}, scene.hpp:69-73), one of
    them pointing at a mesh index that does not exist (skipped on both sides)."""
    rng = np.random.default_rng(13)
    S = pkg.scene
    mats = [pkg.scenes.make_material_textures(rng, 32), pkg.scenes.fallback_textures()]
    sphere = pkg.scenes.uv_sphere(0.6, 24, 12)
    box = pkg.scenes.box(0.8, 0.8, 0.8, 2)
    objs = [(S.translation(-1.5, 0, 0) @ S.rotation_y(30), 0), (S.translation(1.5, 0.3, -1) @ S.scaling(1.5, 0.5, 1.0), 0),
            (S.translation(0, -0.8, 0.5) @ S.rotation_y(-50), 1), (np.eye(4), 7)]
    lights = pkg.scenes.random_lights(rng, 5, (-3, -2, 0.5), (3, 2, 3))
    check(both(pkg, oracle, hip, 128, 96, 256, mats, [sphere + (0,), box + (1,)], objs, lights, settings=(2, 2.2, 1.0)))


def test_work_item_table_overflow_is_reported_and_recovers(pkg, hip):
    """the rasteriser's work-item table is sized from earlier frames; a frame that needs more is dropped LOUDLY
    (ARCTIC_E_CAPACITY at the next synchronising call) and the next one succeeds with the grown table."""
    sc = pkg.scenes.config3(scale=0.1)
    ref_r = sc.upload(hip.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
    ref = ref_r.render_frame(sc.desc, sc.settings)
    items = max(int(ref_r.stats()[1]), int(ref_r.stats()[3]))
    ref_r.close()
    assert items > 256
    r = sc.upload(hip.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
    r.set_option("item_table_floor", 64)
    with pytest.raises(hip.ArcticError) as e:
        r.render_frame(sc.desc, sc.settings)
    assert e.value.code == -5 and "overflow" in str(e.value)
    img = r.render_frame(sc.desc, sc.settings)     # the table has grown to 4x what that frame asked for
    np.testing.assert_array_equal(img, ref)
    np.testing.assert_array_equal(r.render_frame(sc.desc, sc.settings), ref)
    r.close()


@pytest.mark.parametrize("n_mat,n_lights", [(1, 2048), (400, 2048), (400, 40)], ids=["2048-lights", "400-materials-2048-lights", "400-materials-40-lights"])
def test_large_light_and_material_tables(pkg, oracle, hip, n_mat, n_lights):
    """large tables: 2048 lights are 48 KiB of light pairs in LDS (the packed loop) or a long scalar loop; hundreds of
    materials mean tiles with several materials each (the waterfall loop over a tile's distinct descriptors)."""
    rng = np.random.default_rng(n_mat * 7 + n_lights)
    w, h = 96, 64
    mats = [tuple(rng.integers(40, 220, (4, 4, 4), dtype=np.uint8) for _ in range(3)) for _ in range(n_mat)]
    for d, n, m in mats:
        n[..., :2] = rng.integers(118, 138, (4, 4, 2)); n[..., 2] = 255
        m[..., 1] = rng.integers(40, 255, (4, 4)); m[..., 2] = 0
    quad = tri_mesh(pkg, [(-1, -1, 0), (1, -1, 0), (1, 1, 0), (-1, 1, 0)])
    idx = np.array([0, 1, 2, 0, 2, 3], np.uint32)
    meshes = [(quad, idx, k) for k in range(n_mat)]
    cols = int(np.ceil(np.sqrt(n_mat)))
    objs = []
    for k in range(n_mat):
        s = pkg.scene.scaling(3.2 / cols, 2.2 / cols, 1.0)
        t = pkg.scene.translation(-3.2 + (k % cols + 0.5) * 6.4 / cols, -2.2 + (k // cols + 0.5) * 4.4 / cols, 0.0)
        objs.append((t @ s, k))
    lights = pkg.scene.make_lights(rng.uniform((-3, -2, 0.3), (3, 2, 2.5), (n_lights, 3)), rng.uniform(0.0, 0.02, (n_lights, 3)))
    outs = []
    for cls in (oracle.Oracle, hip.Renderer):
        r = cls(w, h, 0, n_lights)
        for m in mats:
            r.create_material(*m)
        for v, i, mat in meshes:
            r.create_mesh(v, i, mat)
        r.update_lights(lights)
        if cls is hip.Renderer:
            r.set_option("keep_float_output", 1)
        desc = camera_scene(pkg, pkg.scene.make_objects(objs), w, h, lights)
        r.render_frame(desc, (0, 2.2, 1.0))
        outs.append(r.read_output()[0].copy())
        if cls is hip.Renderer:
            for path in (1, 2):
                r.set_option("light_path", path)
                r.render_frame(desc, (0, 2.2, 1.0))
                assert np.abs(r.read_output()[0] - outs[-1]).max() <= 3e-6
        r.close()
    assert outs[0].max() > 0.2
    assert np.abs(outs[0] - outs[1]).max() <= TOL


def test_record_table_overflow_is_reported_and_recovers(pkg, oracle, hip):
    """the setup-record table starts at 2 records per source triangle; a scene whose triangles are nearly all clipped into
    three (one vertex behind the camera, one beyond the far plane) overflows it: loud error, then the worst-case table."""
    rng = np.random.default_rng(5)
    n = 6000
    pts = np.zeros((n, 3, 3), np.float32)
    cx, cy = rng.uniform(-1.5, 1.5, n), rng.uniform(-1.0, 1.0, n)
    pts[:, 0] = np.stack([cx - 0.2, cy - 0.2, np.full(n, 9.0)], -1)        # behind the eye (z = 4, looking down -z)
    pts[:, 1] = np.stack([cx + 0.3, cy - 0.1, rng.uniform(-3, 1, n)], -1)  # inside the frustum
    pts[:, 2] = np.stack([cx * 40, cy * 40 + 30, np.full(n, -400.0)], -1)  # beyond the far plane (z_far = 100)
    v = tri_mesh(pkg, pts.reshape(-1, 3))
    mats = [pkg.scenes.fallback_textures()]
    desc = camera_scene(pkg, pkg.scene.make_objects([(np.eye(4, dtype=np.float32), 0)]), 160, 96)
    # make every triangle face the camera: the orientation of a (clipped) triangle on screen is the sign of det[x y w] of its
    # clip-space vertices; flip the index order where it differs, and let the oracle say which sign is "front"
    pv = np.asarray(oracle.frame_constants(desc)[0], np.float64).reshape(4, 4).T          # math matrix from glm memory order
    clip = np.concatenate([pts.astype(np.float64), np.ones((n, 3, 1))], -1) @ pv.T
    sign = np.sign(np.linalg.det(clip[..., [0, 1, 3]]))
    best = None
    for front in (1.0, -1.0):
        order = np.where((sign == front)[:, None], np.array([0, 1, 2])[None, :], np.array([0, 2, 1])[None, :])
        idx = (np.arange(n)[:, None] * 3 + order).reshape(-1).astype(np.uint32)
        o = oracle.Oracle(160, 96, 0, 16); o.create_material(*mats[0]); o.create_mesh(v, idx, 0)
        ref = o.render_frame(desc, (0, 2.2, 1.0))
        if best is None or int(o.stats()[0]) > best[0]:
            best = (int(o.stats()[0]), idx, ref)
        o.close()
    n_recs, idx, ref = best
    assert n_recs > 2 * n + 4096, f"the test scene does not overflow the record table ({n_recs} records)"
    r = hip.Renderer(160, 96, 0, 16); r.create_material(*mats[0]); r.create_mesh(v, idx, 0)
    with pytest.raises(hip.ArcticError) as e:
        r.render_frame(desc, (0, 2.2, 1.0))
    assert e.value.code == -5
    img = r.render_frame(desc, (0, 2.2, 1.0))
    assert int(r.stats()[0]) == n_recs
    d = np.abs(img.astype(np.int16) - ref.astype(np.int16))
    assert d.max() <= 1 and (d != 0).mean() < 2e-3 and (img[..., :3].sum(-1) > 0).mean() > 0.2
    r.close()


def test_profiler_markers_option(pkg, hip):
    """ARCTIC_OPT_MARKERS: roctx ranges named like the reference's Tracy zones; rendering is unchanged."""
    sc = pkg.scenes.config2(scale=0.1)
    r = sc.upload(hip.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
    a = r.render_frame(sc.desc, sc.settings)
    r.set_option("markers", 1)
    b = r.render_frame(sc.desc, sc.settings)
    r.set_option("markers", 0)
    np.testing.assert_array_equal(a, b)
    r.close()


@pytest.mark.parametrize("scale", [0.05, 0.13])
def test_tiles_per_wave_is_placement_only(pkg, hip, scale):
    """ARCTIC_OPT_TILES_PER_WAVE (the library shades one tile per wave below ~3 Mpx, two above): a wave that shades T tiles
    1 / T-th of the frame apart writes the same bytes whatever T, also when the groups of 8 tile rows do not divide by T --
    through the G-buffer pass and through whole frames from the visibility plane."""
    sc = pkg.scenes.config3(scale=scale)
    outs = []
    for T in (0, 1, 2, 3, 5):
        r = sc.upload(hip.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
        r.set_option("tiles_per_wave", T)
        frame = r.render_frame(sc.desc, sc.settings).copy()
        r.pass_shadow_map(sc.desc); r.pass_gbuffer(sc.desc); r.pass_shade(sc.desc, sc.settings)
        outs.append((frame, r.read_output(want=("rgba8",))[2].copy()))
        r.close()
    for frame, passed in outs[1:]:
        np.testing.assert_array_equal(frame, outs[0][0])
        np.testing.assert_array_equal(passed, outs[0][1])
    np.testing.assert_array_equal(outs[0][0], outs[0][1])


def test_resize_across_the_librarys_thresholds(pkg, hip):
    """Renderer::resize (renderer.hpp:116): one handle taken through sizes on either side of the library's own switches -- block
    owners in the forward raster from 4 Mpx, three frames in flight and one tile per wave below 3 Mpx -- renders, after every
    resize, what a fresh handle of that size renders; frames are enqueued back to back (in flight) before each comparison."""
    sc = pkg.scenes.config3(scale=0.25)
    sizes = [(960, 536), (2560, 1600), (640, 360), (2048, 1200), (960, 536)]      # 0.5, 4.1, 0.2, 2.5, 0.5 Mpx
    r = sc.upload(hip.Renderer(*sizes[0], sc.shadow_size, sc.max_lights))
    for w, h in sizes:
        r.resize(w, h)
        d = copy.deepcopy(sc.desc)
        d.camera["aspect"] = w / h
        for k in range(4):
            d.camera["rotation"] = (-15.0 + k, 7.0 * k)
            img = r.render_frame(d, sc.settings)
        fresh = sc.upload(hip.Renderer(w, h, sc.shadow_size, sc.max_lights))
        np.testing.assert_array_equal(img, fresh.render_frame(d, sc.settings))
        for a, b in zip(r.read_gbuffer(), fresh.read_gbuffer()):
            np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))
        fresh.close()
    r.close()
