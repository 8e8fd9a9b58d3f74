"""Known-answer tests that pin the CPU oracle (CPU only).

The reference holds no golden vectors for this path ("parity
unpinned"), so the oracle is pinned by hand-derivable cases K1..K10 of
SURVEY.md 8(c).  Each expected value is re-derived HERE in float64 numpy
straight from the HLSL text (shaders/forward.hlsl:126-193,
shaders/post_process.hlsl:15-57) -- an implementation independent of
oracle/arctic_oracle.cpp -- and the literal numbers of the SURVEY table are
asserted as well.
"""
import numpy as np
import pytest

PI = 3.14159265  # forward.hlsl:1


def unit(v):
    v = np.asarray(v, np.float64)
    return v / np.linalg.norm(v)


def radiance64(n, wo, wi, Li, base, metal, rough):
    """forward.hlsl:126-193 in float64."""
    n, wo, wi, Li, base = (np.asarray(x, np.float64) for x in (n, wo, wi, Li, base))
    h = unit(wo + wi)
    F0 = 0.04 + (base - 0.04) * metal
    F = F0 + (1 - F0) * np.clip(1 - max(h @ wo, 0), 0, 1) ** 5
    a2 = (rough * rough) ** 2
    ndh = max(n @ h, 0)
    d = ndh * ndh * (a2 - 1) + 1
    NDF = a2 / (PI * d * d)
    k = (rough + 1) ** 2 / 8
    g = lambda x: x / (x * (1 - k) + k)
    ndwo, ndwi = max(n @ wo, 0), max(n @ wi, 0)
    spec = NDF * g(ndwo) * g(ndwi) * F / (4 * ndwo * ndwi + 0.0001)
    kD = (1 - F) * (1 - metal)
    return (kD * base / PI + spec) * Li * ndwi


def tonemap64(method, c, gamma=2.2, exposure=1.0):
    c = np.asarray(c, np.float64)
    if method == 1:
        t = 1 - np.exp(-c * exposure)
    elif method == 2:
        mi = np.array([[0.59719, 0.35458, 0.04823], [0.07600, 0.90834, 0.01566], [0.02840, 0.13383, 0.837]])
        mo = np.array([[1.60475, -0.53108, -0.07367], [-0.10208, 1.10813, -0.00605], [-0.00327, -0.07276, 1.07]])
        v = mi @ c
        v = (v * (v + 0.0245786) - 0.000090537) / (v * (0.983729 * v + 0.4329510) + 0.238081)
        t = np.clip(mo @ v, 0, 1)
    else:
        t = c / (c + 1)
    return t, np.abs(t) ** (1 / gamma)


K_RADIANCE = [
    # name, n, wo, wi, Li, base, metal, rough, SURVEY value
    ("K1", (0, 0, 1), (0, 0, 1), (0, 0, 1), (1, 1, 1), (1, 1, 1), 0.0, 1.0, (0.30876051, 0.30876051, 0.30876051)),
    ("K2", (0, 0, 1), (0, 0, 1), unit((1, 0, 1)), (8, 8, 8), (1, 0.5, 0.25), 1.0, 0.5, (0.89273681, 0.44636955, 0.22318591)),
    ("K3", (0, 0, 1), unit((0, 1, 1)), (1, 0, 0), (3, 2, 1), (0.7, 0.6, 0.5), 0.3, 0.4, (0, 0, 0)),
]


@pytest.mark.parametrize("case", K_RADIANCE, ids=[c[0] for c in K_RADIANCE])
def test_outgoing_radiance_kat(oracle, case):
    _, n, wo, wi, Li, base, metal, rough, survey = case
    want = radiance64(n, wo, wi, Li, base, metal, rough)
    np.testing.assert_allclose(want, survey, rtol=2e-6, atol=1e-9)
    got = oracle.outgoing_radiance(n, wo, wi, Li, base, metal, rough)
    np.testing.assert_allclose(got, want, rtol=3e-6, atol=1e-9)


def test_k4_default_point_light(oracle):
    # src/app.hpp:57-60: light at (0,1,0), colour (10,0,0); surface at the origin, n = wo = +y
    d = np.array([0.0, 1.0, 0.0])
    dist = np.linalg.norm(d)
    want = radiance64((0, 1, 0), (0, 1, 0), d / dist, np.array([10.0, 0, 0]) / dist**2, (0.8,) * 3, 0.0, 0.5)
    np.testing.assert_allclose(want, (2.95390302, 0, 0), rtol=2e-6)
    got = oracle.outgoing_radiance((0, 1, 0), (0, 1, 0), (0, 1, 0), (10, 0, 0), (0.8,) * 3, 0.0, 0.5)
    np.testing.assert_allclose(got, want, rtol=3e-6)


K_TONEMAP = [
    ("K5", 0, (0.15254237, 0.5, 0.8), (0.42541604, 0.72974005, 0.90354543)),
    ("K6", 1, (0.16472979, 0.63212056, 0.98168436), None),
    ("K7", 2, (0.36484199, 0.62692476, 0.90948523), (0.63234885, 0.8087703, 0.957791)),
]


@pytest.mark.parametrize("case", K_TONEMAP, ids=[c[0] for c in K_TONEMAP])
def test_tonemap_kat(oracle, case):
    _, method, survey_tm, survey_out = case
    c = (0.18, 1.0, 4.0)
    tm64, out64 = tonemap64(method, c)
    np.testing.assert_allclose(tm64, survey_tm, rtol=3e-6)
    if survey_out is not None:
        np.testing.assert_allclose(out64, survey_out, rtol=3e-6)
    tm, out = oracle.tonemap(method, 2.2, 1.0, c)
    np.testing.assert_allclose(tm, tm64, rtol=3e-6)
    np.testing.assert_allclose(out, out64, rtol=3e-6)


def test_tonemap_unknown_method_is_reinhard(oracle):
    # post_process.hlsl:76-81: `default:` shares the Reinhard case
    c = (0.3, 2.0, 9.0)
    for m in (3, 7, -1):
        np.testing.assert_array_equal(oracle.tonemap(m, 2.2, 1.0, c)[1], oracle.tonemap(0, 2.2, 1.0, c)[1])


def test_k8_dir_from_rot_default_sun(oracle):
    # scene.cpp:9-19 with the default sun rotation of src/app.hpp:53
    x, y = np.deg2rad(-70.0), np.deg2rad(12.0)
    want = np.array([np.cos(x) * np.cos(y), np.sin(x), np.cos(x) * np.sin(y)])
    np.testing.assert_allclose(want, (0.33454618, -0.93969262, 0.07110999), rtol=2e-6)
    np.testing.assert_allclose(oracle.dir_from_rot((-70.0, 12.0)), want, rtol=1e-6)
    np.testing.assert_allclose(oracle.dir_from_rot((0.0, 0.0)), (1, 0, 0), atol=1e-7)


def test_k9_fallback_material(oracle, pkg):
    # assets/white.png / normal.png / white.png (src/app.cpp:194-245): base 1, metal 1, rough 1, flat normal
    o = oracle.Oracle(8, 8)
    o.create_material(*pkg.scenes.fallback_textures())
    for u, v in ((0.1, 0.2), (0.77, 0.5), (3.4, -2.25)):
        s = o.fetch_surface(0, u, v)
        np.testing.assert_allclose(s[0:3], 1.0, rtol=1e-6)                                       # sRGB 255 -> 1.0
        np.testing.assert_allclose(s[3:6], (128 / 255 * 2 - 1, (1 - 128 / 255) * 2 - 1, 1.0), atol=2e-7)
        np.testing.assert_allclose(s[6:9], unit((0.003922, -0.003922, 1.0)), atol=1e-5)
        assert s[9] == pytest.approx(1.0) and s[10] == pytest.approx(1.0)


def test_k10_shadow_edge_cases(oracle):
    S = 64
    m = np.full((S, S), 0.4, np.float32)
    assert oracle.calculate_shadow(m, (0, 0, 1.0001, 1)) == 0.0     # z > 1 -> lit
    assert oracle.calculate_shadow(m, (1.2, 0, 0.5, 1)) == 0.0      # outside the light frustum -> lit
    assert oracle.calculate_shadow(m, (0, 0, 0.5, 1)) == 1.0        # behind the occluder
    assert oracle.calculate_shadow(m, (0, 0, 0.3, 1)) == 0.0        # in front of it
    assert oracle.calculate_shadow(m, (0, 0, -0.2, 1)) == 0.0       # no z < 0 test in the HLSL: still sampled, lit
    assert oracle.calculate_shadow(None, (0, 0, 0.5, 1)) == 0.0     # no shadow map (config 1)


def test_shadow_pcf_fraction_and_bilinear_compare(oracle):
    # forward.hlsl:84-92: 25 taps spaced 1e-4 in uv, each compared against BILINEARLY FILTERED depth
    S = 4000
    m = np.ones((S, S), np.float32)
    m[:, : S // 2] = 0.2    # left half occluder at depth 0.2, right half clear
    # receiver exactly at the texel boundary column: uv.x = 0.5 -> taps straddle the edge
    ls = (0.0, 0.0, 0.5, 1.0)      # ndc (0,0) -> uv (0.5,0.5)
    got = oracle.calculate_shadow(m, ls)
    # float64 restatement
    cnt = 0
    for i in range(-2, 3):
        u = 0.5 + i * 1e-4
        x = u * S - 0.5
        x0 = int(np.floor(x))
        fx = x - x0
        d = m[100, x0] * (1 - fx) + m[100, x0 + 1] * fx
        cnt += 5 * (0.5 > d)
    assert got == pytest.approx(cnt / 25.0)
    assert 0.0 < got < 1.0


def test_unorm8_store_rule(oracle):
    # D3D float->UNORM: saturate, *255, +0.5, truncate; NaN -> 0
    assert oracle.to_unorm8(0.0) == 0 and oracle.to_unorm8(1.0) == 255 and oracle.to_unorm8(7.0) == 255
    assert oracle.to_unorm8(-3.0) == 0 and oracle.to_unorm8(float("nan")) == 0
    assert oracle.to_unorm8(0.5) == 128            # 127.5 + 0.5 = 128
    assert oracle.to_unorm8(127.4 / 255) == 127 and oracle.to_unorm8(127.6 / 255) == 128


def test_matrices_hand_derived(oracle, pkg):
    """glm restatement: eye at origin looking down -Z: view = identity."""
    S = pkg.scene
    # rotation (0,-90): dir = (cos0*cos(-90), 0, cos0*sin(-90)) = (0,0,-1)
    desc = S.SceneDesc(camera=dict(eye=(0, 0, 0), rotation=(0.0, -90.0), aspect=2.0, fov_y=90.0, z_near_far=(1.0, 3.0)),
                       ambient=0.1, sun=dict(position=(0, 0, 0), rotation=(0.0, -90.0), color=(1, 1, 1)),
                       objects=np.zeros(0, S.OBJECT_DTYPE))
    pv, lpv, sd = oracle.frame_constants(desc)   # [col][row]
    t = np.tan(np.deg2rad(45.0))
    want = np.zeros((4, 4))
    want[0][0] = 1 / (2.0 * t)
    want[1][1] = 1 / t
    want[2][2] = 3.0 / (1.0 - 3.0)
    want[2][3] = -1
    want[3][2] = -(3.0 * 1.0) / (3.0 - 1.0)
    np.testing.assert_allclose(pv, want, atol=1e-6)
    # a point on the near plane maps to depth 0, on the far plane to depth 1
    for z, d in ((-1.0, 0.0), (-3.0, 1.0)):
        clip = pv.T @ np.array([0, 0, z, 1.0])
        assert clip[2] / clip[3] == pytest.approx(d, abs=1e-6)
    # ortho +-16, z 0.1..50 (scene.cpp:68)
    wo = np.zeros((4, 4))
    wo[0][0] = wo[1][1] = 2 / 32.0
    wo[2][2] = -1 / 49.9
    wo[3][2] = -0.1 / 49.9
    wo[3][3] = 1
    np.testing.assert_allclose(lpv, wo, atol=1e-6)
    np.testing.assert_allclose(sd, (0, 0, -1), atol=1e-6)


def test_look_at_general(oracle, pkg):
    """lookAtRH against a float64 restatement for the reference's default camera/sun (src/app.hpp:42-55)."""
    S = pkg.scene
    desc = S.SceneDesc(camera=dict(eye=(0, 5, 0), rotation=(-15.0, 30.0), aspect=16 / 9, fov_y=45.0, z_near_far=(0.1, 1000.0)),
                       ambient=0.1, sun=pkg.scenes.DEFAULT_SUN, objects=np.zeros(0, S.OBJECT_DTYPE))
    pv, lpv, sd = oracle.frame_constants(desc)

    def look(eye, rot):
        x, y = np.deg2rad(rot)
        f = np.array([np.cos(x) * np.cos(y), np.sin(x), np.cos(x) * np.sin(y)])
        s = unit(np.cross(f, (0, 1, 0)))
        u = np.cross(s, f)
        m = np.eye(4)
        m[0, :3], m[1, :3], m[2, :3] = s, u, -f
        m[:3, 3] = -(s @ eye), -(u @ eye), f @ eye
        return m   # math convention [row][col]

    t = np.tan(np.deg2rad(45.0) / 2)
    proj = np.zeros((4, 4))
    proj[0, 0], proj[1, 1], proj[2, 2], proj[3, 2], proj[2, 3] = 1 / (16 / 9 * t), 1 / t, 1000 / (0.1 - 1000), -1, -(1000 * 0.1) / (1000 - 0.1)
    np.testing.assert_allclose(pv.T, proj @ look(np.array([0, 5.0, 0]), (-15.0, 30.0)), rtol=2e-5, atol=2e-5)
    orth = np.diag([2 / 32, 2 / 32, -1 / 49.9, 1.0])
    orth[2, 3] = -0.1 / 49.9
    np.testing.assert_allclose(lpv.T, orth @ look(np.array([-10, 32, -2.48]), (-70.0, 12.0)), rtol=2e-5, atol=2e-5)


def test_binary16_rounding_of_the_hdr_target(oracle):
    """the reference's colour target is R16G16B16A16_FLOAT (forward_pass.cpp:149): round-to-nearest-even to binary16."""
    rng = np.random.default_rng(4)
    xs = np.concatenate([np.abs(rng.standard_normal(3000)).astype(np.float32) * s for s in (1e-8, 1e-5, 1e-3, 1.0, 100.0, 3e4)]
                        + [np.array([0.0, 65504.0, 65519.9, 6e-8, 5.96e-8, 2.98e-8, 2.99e-8, 1.0009766, 1.00048828125], np.float32)])
    got = np.array([oracle.through_half(float(x)) for x in xs], np.float32)
    np.testing.assert_array_equal(got, xs.astype(np.float16).astype(np.float32))
    assert np.isinf(oracle.through_half(65520.0))
