"""Skybox restatement of the oracle (CPU only): skybox.hlsl:61-90 + skybox_pass.cpp:104-138.

The reference draws a unit cube at depth 1 with proj * mat3(lookAtRH) and looks the environment map up along the
interpolated cube position.  The oracle (and the HIP kernel) skip the cube and use the pixel's view ray.  These tests
re-derive, in float64 numpy and independently of oracle/arctic_oracle.cpp, (1) that the two formulations give the same
direction, (2) the equirect uv and (3) the LINEAR/WRAP filter result.  No reference golden image exists: parity unpinned.
"""
import numpy as np
import pytest

CUBE = np.array([  # skybox.hlsl:1-43
    (-1, 1, -1), (-1, -1, -1), (1, -1, -1), (1, -1, -1), (1, 1, -1), (-1, 1, -1),
    (-1, -1, 1), (-1, -1, -1), (-1, 1, -1), (-1, 1, -1), (-1, 1, 1), (-1, -1, 1),
    (1, -1, -1), (1, -1, 1), (1, 1, 1), (1, 1, 1), (1, 1, -1), (1, -1, -1),
    (-1, -1, 1), (-1, 1, 1), (1, 1, 1), (1, 1, 1), (1, -1, 1), (-1, -1, 1),
    (-1, 1, -1), (1, 1, -1), (1, 1, 1), (1, 1, 1), (-1, 1, 1), (-1, 1, -1),
    (-1, -1, -1), (-1, -1, 1), (1, -1, -1), (1, -1, -1), (-1, -1, 1), (1, -1, 1)], np.float64)

CAMERAS = [dict(eye=(0.0, 0.0, 5.0), rotation=(0.0, -90.0), aspect=16 / 9, fov_y=45.0, z_near_far=(0.1, 100.0)),
           dict(eye=(3.0, 2.0, 1.0), rotation=(-25.0, 140.0), aspect=1.0, fov_y=60.0, z_near_far=(0.1, 100.0)),
           dict(eye=(0.0, 1.0, 0.0), rotation=(40.0, 10.0), aspect=2.0, fov_y=30.0, z_near_far=(0.5, 50.0))]


def proj_view_no_translation(cam):
    """scene.cpp:26-38 in float64."""
    xr, yr = np.deg2rad(cam["rotation"])
    f = np.array([np.cos(xr) * np.cos(yr), np.sin(xr), np.cos(xr) * np.sin(yr)])
    f /= np.linalg.norm(f)
    s = np.cross(f, (0, 1, 0)); s /= np.linalg.norm(s)
    u = np.cross(s, f)
    view = np.eye(4); view[0, :3], view[1, :3], view[2, :3] = s, u, -f
    t = np.tan(np.deg2rad(cam["fov_y"]) / 2)
    zn, zf = cam["z_near_far"]
    proj = np.zeros((4, 4))
    proj[0, 0], proj[1, 1], proj[2, 2], proj[3, 2], proj[2, 3] = 1 / (cam["aspect"] * t), 1 / t, zf / (zn - zf), -1, -(zf * zn) / (zf - zn)
    return proj @ view


def cube_direction(cam, px, py, W, H):
    """what the reference's rasteriser hands ps_main at pixel (px, py): the perspective-correct interpolation of the cube
    vertex positions over the (clipped-or-not) triangle covering the pixel."""
    pv = proj_view_no_translation(cam)
    ndc = np.array([(px + 0.5) * 2 / W - 1, 1 - (py + 0.5) * 2 / H])
    for t in range(12):
        v = CUBE[3 * t:3 * t + 3]
        clip = (pv @ np.c_[v, np.ones(3)].T).T
        # solve for barycentrics in clip space: sum b_i * clip_i.xy = ndc * sum b_i * clip_i.w  (exact, handles w<=0 corners)
        A = np.array([clip[:, 0] - ndc[0] * clip[:, 3], clip[:, 1] - ndc[1] * clip[:, 3], np.ones(3)])
        try:
            b = np.linalg.solve(A, np.array([0, 0, 1.0]))
        except np.linalg.LinAlgError:
            continue
        if (b >= -1e-12).all() and (b @ clip[:, 3]) > 0:
            return b @ v
    raise AssertionError("no cube face covers the pixel")


@pytest.mark.parametrize("cam", CAMERAS, ids=["front", "oblique", "up"])
def test_view_ray_equals_interpolated_cube_position(oracle, cam):
    W, H = 64, 48
    o = oracle.Oracle(W, H)
    for px, py in [(0, 0), (63, 0), (0, 47), (63, 47), (31, 23), (32, 24), (10, 40), (50, 5)]:
        d = o.sky_ray(cam, px, py).astype(np.float64)
        c = cube_direction(cam, px, py, W, H)
        assert np.abs(d / np.linalg.norm(d) - c / np.linalg.norm(c)).max() < 2e-6, (px, py)
    o.close()


def equirect_uv(d):
    d = np.asarray(d, np.float64); d = d / np.linalg.norm(d)
    return np.array([np.arctan2(d[2], d[0]) * np.float64(np.float32(0.1591)) + 0.5,
                     -(np.arcsin(d[1]) * np.float64(np.float32(0.3183)) + 0.5)])


def bilinear_wrap(img, uv):
    """D3D LINEAR + WRAP: texel centres at (i + 0.5) / n."""
    h, w = img.shape[:2]
    x, y = (uv[0] % 1.0) * w - 0.5, (uv[1] % 1.0) * h - 0.5
    x0, y0 = int(np.floor(x)), int(np.floor(y))
    fx, fy = np.float64(np.float32(x - x0)), np.float64(np.float32(y - y0))
    p = lambda yy, xx: img[yy % h, xx % w, :3].astype(np.float64)
    return (p(y0, x0) * (1 - fx) + p(y0, x0 + 1) * fx) * (1 - fy) + (p(y0 + 1, x0) * (1 - fx) + p(y0 + 1, x0 + 1) * fx) * fy


def test_equirect_uv_known_answers(oracle):
    o = oracle.Oracle(8, 8)
    for d, want in [((1, 0, 0), (0.5, -0.5)),                                   # +x: centre of the map
                    ((0, 0, 1), (np.pi / 2 * 0.1591 + 0.5, -0.5)),              # +z: a quarter turn
                    ((0, 1, 0), (0.5, -(np.pi / 2 * 0.3183 + 0.5))),            # zenith: v = -0.99998 (wraps to the top row)
                    ((0, -1, 0), (0.5, -(0.5 - np.pi / 2 * 0.3183)))]:
        _, uv = o.sample_environment(d)
        assert np.abs(uv - want).max() < 1e-7, d
    rng = np.random.default_rng(5)
    for d in rng.standard_normal((50, 3)):
        _, uv = o.sample_environment(d)
        assert np.abs(uv - equirect_uv(np.float32(d))).max() < 1e-12
    o.close()


def test_environment_filter(oracle, pkg):
    env = pkg.scenes.synthetic_hdri(64, 32)
    assert env[..., :3].max() > 20 and env[..., :3].min() >= 0   # HDR contrast is present
    o = oracle.Oracle(8, 8)
    o.create_hdri(env)
    rng = np.random.default_rng(6)
    dirs = list(rng.standard_normal((200, 3))) + [(-1, 0, 1e-9), (-1, 0, -1e-9), (0, 1, 0), (0, -1, 0), (1, 0, 0)]   # incl. the u seam and poles
    for d in dirs:
        d = np.float32(d)
        rgb, uv = o.sample_environment(d)
        want = bilinear_wrap(env, equirect_uv(d))
        assert np.abs(rgb - want).max() <= 1e-6 * max(1.0, np.abs(want).max()), d
    # a constant map is reproduced exactly whatever the weights
    o.create_hdri(np.full((4, 8, 4), 0.75, np.float32))
    for d in dirs[:20]:
        assert np.abs(o.sample_environment(np.float32(d))[0] - 0.75).max() < 1e-7
    o.close()


def test_sky_fills_exactly_the_uncovered_pixels(oracle, pkg):
    sc = pkg.scenes.config1(scale=0.125)
    o = sc.upload(oracle.Oracle(sc.width, sc.height, sc.shadow_size, sc.max_lights))
    o.pass_gbuffer(sc.desc)
    o.pass_shade(sc.desc, sc.settings)
    black = o.read_output()[1].copy()
    sc.environment = pkg.scenes.synthetic_hdri(128, 64)
    o.create_hdri(sc.environment)
    o.pass_shade(sc.desc, sc.settings)
    hdr = o.read_output()[1]
    covered = o.read_gbuffer()[1] != 0xFFFFFFFF
    assert 0.05 < covered.mean() < 0.95
    np.testing.assert_array_equal(hdr[covered], black[covered])          # geometry wins (depth LESS_EQUAL against z = w)
    assert (black[~covered] == 0).all() and (hdr[~covered].sum(-1) > 0).all()
    # the row-sharded oracle sees the same sky (global pixel rows)
    half = sc.height // 2
    o2 = sc.upload(oracle.Oracle(sc.width, sc.height, sc.shadow_size, sc.max_lights, row_begin=half, row_end=sc.height))
    o2.pass_gbuffer(sc.desc)
    o2.pass_shade(sc.desc, sc.settings)
    np.testing.assert_array_equal(o2.read_output()[1], hdr[half:])
    o.close(); o2.close()
