"""Rasterisation rules of the oracle (CPU): what D3D12 does in fixed function for the reference
(forward_pass.cpp:137-151, shadow_map_pass.cpp:96-97) and what the HIP prepass must reproduce bit for bit."""
import numpy as np
import pytest


def flat_material(pkg):
    return pkg.scenes.fallback_textures()


def ortho_scene(pkg, objects, aspect=1.0):
    """camera at +z looking down -z with a 90 degree fov: at z = 0 the view spans [-d, d] for eye distance d."""
    S = pkg.scene
    return S.SceneDesc(camera=dict(eye=(0, 0, 4.0), rotation=(0.0, -90.0), aspect=aspect, fov_y=90.0, z_near_far=(0.1, 100.0)),
                       ambient=0.1, sun=dict(position=(0, 10, 0), rotation=(-90.0, 0.0), color=(1, 1, 1)), objects=objects)


def tri_mesh(pkg, pts, uv=None):
    S = pkg.scene
    v = np.zeros(len(pts), S.VERTEX_DTYPE)
    v["position"] = pts
    v["normal"] = (0, 0, 1)
    v["tangent"] = (1, 0, 0)
    v["bitangent"] = (0, 1, 0)
    v["tex_coords"] = uv if uv is not None else np.asarray(pts)[:, :2]
    return v


def render_cov(oracle, pkg, verts, idx, size=32):
    o = oracle.Oracle(size, size)
    o.create_material(*flat_material(pkg))
    o.create_mesh(tri_mesh(pkg, verts), idx, 0)
    desc = ortho_scene(pkg, pkg.scene.make_objects([(np.eye(4), 0)]))
    o.pass_gbuffer(desc)
    return o, o.read_gbuffer()


def test_front_faces_are_counter_clockwise(oracle, pkg):
    ccw = [(-2, -2, 0), (2, -2, 0), (0, 2, 0)]          # counter-clockwise seen from +z (the camera side)
    _, (_, mat, _, _) = render_cov(oracle, pkg, ccw, [0, 1, 2])
    assert (mat != 0xFFFFFFFF).sum() > 50
    _, (_, mat, _, _) = render_cov(oracle, pkg, ccw, [0, 2, 1])   # clockwise: back face, culled
    assert (mat != 0xFFFFFFFF).sum() == 0


def test_shared_edge_is_covered_exactly_once(oracle, pkg):
    """top-left rule: two triangles sharing an edge never both cover a pixel and leave no gap."""
    rng = np.random.default_rng(3)
    for _ in range(20):
        q = rng.uniform(-3.5, 3.5, (4, 2))
        # order the quad counter-clockwise around its centroid
        c = q.mean(0)
        q = q[np.argsort(np.arctan2(q[:, 1] - c[1], q[:, 0] - c[0]))]
        pts = [(x, y, 0.0) for x, y in q]
        counts = np.zeros((48, 48), int)
        for tri in ([0, 1, 2], [0, 2, 3]):
            _, (_, mat, _, _) = render_cov(oracle, pkg, pts, tri, size=48)
            counts += (mat != 0xFFFFFFFF)
        _, (_, both, _, _) = render_cov(oracle, pkg, pts, [0, 1, 2, 0, 2, 3], size=48)
        assert counts.max() <= 1, "a pixel on the shared edge was rasterised by both triangles"
        np.testing.assert_array_equal(counts == 1, both != 0xFFFFFFFF)


def test_pixel_centres_at_half(oracle, pkg):
    """a 2x2-pixel square aligned to the pixel grid covers exactly the 4 pixels whose centres it contains."""
    size = 32                                    # eye distance 4, fov 90 -> the view spans [-4,4]: 1 pixel = 0.25
    x0, x1 = -4 + 10 * 0.25, -4 + 12 * 0.25      # pixel columns 10, 11
    y1, y0 = 4 - 5 * 0.25, 4 - 7 * 0.25          # pixel rows 5, 6 (y down)
    pts = [(x0, y0, 0), (x1, y0, 0), (x1, y1, 0), (x0, y1, 0)]
    _, (_, mat, _, _) = render_cov(oracle, pkg, pts, [0, 1, 2, 0, 2, 3], size=size)
    ys, xs = np.nonzero(mat != 0xFFFFFFFF)
    assert sorted(zip(ys.tolist(), xs.tolist())) == [(5, 10), (5, 11), (6, 10), (6, 11)]


def test_depth_less_first_drawn_wins_ties(oracle, pkg):
    a = [(-2, -2, 0), (2, -2, 0), (0, 2, 0)]
    o = oracle.Oracle(32, 32)
    o.create_material(*flat_material(pkg))
    o.create_material(*flat_material(pkg))
    o.create_mesh(tri_mesh(pkg, a), [0, 1, 2], 0)
    o.create_mesh(tri_mesh(pkg, a), [0, 1, 2], 1)            # identical geometry, second material
    near = [(x, y, 1.0) for x, y, _ in a]
    o.create_mesh(tri_mesh(pkg, near), [0, 1, 2], 1)
    S = pkg.scene
    o.pass_gbuffer(ortho_scene(pkg, S.make_objects([(np.eye(4), 0), (np.eye(4), 1)])))
    mat = o.read_gbuffer()[1]
    assert set(np.unique(mat)) == {0, 0xFFFFFFFF}            # equal depth: LESS keeps the first drawn
    o.pass_gbuffer(ortho_scene(pkg, S.make_objects([(np.eye(4), 0), (np.eye(4), 2)])))
    mat = o.read_gbuffer()[1]
    assert 1 in np.unique(mat)                               # nearer triangle wins where it covers


def test_perspective_correct_interpolation(oracle, pkg):
    """a quad receding in depth: interpolated world position must equal the analytic ray/plane hit."""
    pts = [(-3, -1.5, 0.0), (3, -1.5, 0.0), (3, -1.0, -30.0), (-3, -1.0, -30.0)]
    o, (attrs, mat, depth, _) = render_cov(oracle, pkg, pts, [0, 1, 2, 0, 2, 3], size=64)
    ys, xs = np.nonzero(mat != 0xFFFFFFFF)
    assert len(ys) > 100
    eye = np.array([0, 0, 4.0])
    n = np.cross(np.subtract(pts[1], pts[0]), np.subtract(pts[3], pts[0]))
    for y, x in list(zip(ys, xs))[::17]:
        ndc = np.array([(x + 0.5) / 32 - 1, 1 - (y + 0.5) / 32])
        d = np.array([ndc[0], ndc[1], -1.0])                 # fov 90, aspect 1
        t = np.dot(np.subtract(pts[0], eye), n) / np.dot(d, n)
        hit = eye + t * d
        np.testing.assert_allclose(attrs[y, x, 11:14], hit, rtol=2e-4, atol=2e-4)
        np.testing.assert_allclose(attrs[y, x, 0:2], hit[:2], rtol=2e-4, atol=2e-4)   # uv = xy of the vertex


def test_near_plane_clipping(oracle, pkg):
    """a triangle passing through the camera plane is clipped, not dropped and not wrapped around."""
    pts = [(-1, -1, 0.0), (1, -1, 0.0), (0, -1, 10.0)]       # a floor triangle running under and behind the eye (eye z = 4)
    o, (attrs, mat, depth, _) = render_cov(oracle, pkg, pts, [0, 2, 1], size=64)   # wound so its upper side faces the eye
    cov = mat != 0xFFFFFFFF
    assert cov.sum() > 100
    assert np.isfinite(attrs[cov]).all()
    assert (attrs[cov][:, 13] <= 4.0 - 0.1 + 1e-3).all()     # every visible point lies beyond the near plane
    assert cov[:32].sum() == 0 and cov[60:].sum() > 0        # nothing above the horizon, and it reaches the bottom edge


def test_shadow_pass_culls_front_faces(oracle, pkg):
    """ShadowMapPass uses CULL_MODE_FRONT: a single-sided quad facing the sun writes nothing, facing away it does."""
    S = pkg.scene
    up = [(-3, 0, 3), (3, 0, 3), (3, 0, -3), (-3, 0, -3)]    # counter-clockwise seen from +y: front face towards the sun
    for idx, expect in (([0, 1, 2, 0, 2, 3], False), ([0, 2, 1, 0, 3, 2], True)):
        o = oracle.Oracle(16, 16, shadow_size=128)
        o.create_material(*flat_material(pkg))
        o.create_mesh(tri_mesh(pkg, up), idx, 0)
        desc = S.SceneDesc(camera=dict(eye=(0, 2, 8), rotation=(0.0, -90.0), aspect=1.0, fov_y=45.0, z_near_far=(0.1, 100.0)),
                           ambient=0.1, sun=dict(position=(0, 20, 0.01), rotation=(-89.9, 0.0), color=(1, 1, 1)),
                           objects=S.make_objects([(np.eye(4), 0)]))
        o.pass_shadow_map(desc)
        assert ((o.read_shadow_map() < 1.0).sum() > 0) == expect
