"""ARCTIC_OPT_SMALL_TRIANGLES (round 5): the shadow pass's set-up kernel draws triangles whose bounding box holds at most 64 pixels itself -- a
lane per pixel of the boxes, the edge functions as 32-bit integers relative to the box, one atomicMin per covered pixel -- instead of writing a
record and 16x16 work items for the item rasteriser (needs an MI355X).

The reference hands every triangle to the hardware rasteriser (src/renderer/shadow_map_pass.cpp:96-97,157-167, shaders/depth.hlsl:7-10): one depth
per texel whatever the path, so the map must be the same bit for bit, and the count of triangles set up (arctic_stats) with it.  Compared: the
option off / on, and both against the oracle; scenes whose sun sees mostly small triangles (configs 2, 3), random triangle soups from sub-texel
slivers to triangles larger than the map (whose scissored boxes can be small while their edge functions are not: the 32-bit bounds must refuse
them), maps of ragged sizes, block owners in the shadow pass, whole frames under a moving sun.  (A scissored map -- a rank's slice of a sharded shadow map --
runs with the path on in tests/test_gpu_exchange_loopback.py: every rank's map against the single-device map.)
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def shadow(hip, sc, small, desc=None, debug=0, owner=-1):
    r = sc.upload(hip.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
    r.set_option("small_triangles", small)
    r.set_option("raster_owner", owner)
    if debug:
        r.set_option("debug", debug)
    r.pass_shadow_map(desc or sc.desc)
    r.pass_shadow_map(desc or sc.desc)     # (the second pass sizes its tables from the first)
    m = r.read_shadow_map().view(np.uint32).copy()
    st = [int(x) for x in r.stats()[:4]]
    r.close()
    return m, st


@pytest.mark.parametrize("cfg,scale", [(2, 0.25), (2, 0.5), (3, 0.2), (3, 0.5)])
def test_small_triangles_change_nothing(pkg, hip, oracle, cfg, scale):
    sc = pkg.scenes.CONFIGS[cfg](scale=scale)
    (m0, s0), (m1, s1) = shadow(hip, sc, 0), shadow(hip, sc, 1)
    assert (m0 != 0x3F800000).mean() > 0.002
    np.testing.assert_array_equal(m0, m1)
    assert s0[2] == s1[2]                      # triangles set up: the small ones still count
    assert s1[3] < 0.8 * s0[3]                 # not vacuous: work items that were never made
    print(f"config {cfg} x{scale}: {s0[2]} shadow records, work items {s0[3]} -> {s1[3]}")
    if scale <= 0.25:
        o = sc.upload(oracle.Oracle(sc.width, sc.height, sc.shadow_size, sc.max_lights))
        o.pass_shadow_map(sc.desc)
        np.testing.assert_array_equal(o.read_shadow_map().view(np.uint32), m1)
        assert int(o.stats()[2]) == s1[2]
        o.close()


def soup(pkg, rng, n, size_lo, size_hi, spread):
    """n random triangles: centres in a box of +-spread metres around the sun's frustum axis region, edge lengths log-uniform in [size_lo, size_hi]"""
    c = rng.uniform((-spread, 0.0, -spread), (spread, 12.0, spread), (n, 1, 3))
    size = np.exp(rng.uniform(np.log(size_lo), np.log(size_hi), (n, 1, 1)))
    pts = (c + rng.normal(size=(n, 3, 3)) * size).astype(np.float32).reshape(-1, 3)
    v = np.zeros(len(pts), pkg.scene.VERTEX_DTYPE)
    v["position"] = pts
    v["normal"], v["tangent"], v["bitangent"] = (0, 1, 0), (1, 0, 0), (0, 0, 1)
    v["tex_coords"] = pts[:, :2]
    return v, np.arange(len(pts), dtype=np.uint32)


@pytest.mark.parametrize("S,n,lo,hi,spread", [(512, 20000, 0.005, 0.5, 18.0), (1000, 30000, 0.002, 40.0, 24.0), (97, 4000, 0.01, 3.0, 17.0),
                                             (2048, 60000, 0.01, 0.2, 16.0), (4000, 20000, 0.004, 6.0, 30.0)],
                         ids=["slivers-512", "any-size-1000", "ragged-97", "dense-2048", "huge-4000"])
def test_triangle_soup(pkg, hip, oracle, S, n, lo, hi, spread):
    rng = np.random.default_rng(S * 7 + n)
    mesh = soup(pkg, rng, n, lo, hi, spread)
    mats = [pkg.scenes.fallback_textures()]
    desc = pkg.scene.SceneDesc(camera=dict(eye=(0, 5, 0), rotation=(-15.0, 0.0), aspect=2.0, fov_y=45.0, z_near_far=(0.1, 100.0)), ambient=0.1,
                               sun=pkg.scenes.DEFAULT_SUN, objects=pkg.scene.make_objects([(np.eye(4, dtype=np.float32), 0)]))
    sc = pkg.scenes.SyntheticScene("soup", 64, 32, S, 16, mats, [mesh + (0,)], desc, np.zeros(0, pkg.scene.LIGHT_DTYPE), (0, 2.2, 1.0))
    (m0, s0), (m1, s1) = shadow(hip, sc, 0), shadow(hip, sc, 1)
    assert (m0 != 0x3F800000).mean() > 0.002
    np.testing.assert_array_equal(m0, m1)
    assert s0[2] == s1[2] and s1[3] <= s0[3]
    (mi, si) = shadow(hip, sc, 1, debug=32)    # the integer path forced: the small path stands back (the switch is an A/B of the item rasterisers)
    np.testing.assert_array_equal(m0, mi)
    assert si[3] >= s0[3]                      # every triangle became work items again (the integer path has no reachability masks: a few more than off)
    (mo, so) = shadow(hip, sc, 1, owner=2)     # block owners in the shadow pass MERGE into the map the set-up kernel has drawn the small triangles into; bins that overflow
    np.testing.assert_array_equal(m0, mo)      # (dense soups) leave their items to the atomic rasteriser behind them
    assert so[2] == s0[2] and so[3] == s1[3]
    if S <= 1000:
        o = sc.upload(oracle.Oracle(sc.width, sc.height, sc.shadow_size, sc.max_lights))
        o.pass_shadow_map(sc.desc)
        np.testing.assert_array_equal(o.read_shadow_map().view(np.uint32), m1)
        assert int(o.stats()[2]) == s1[2]
        o.close()


@pytest.mark.parametrize("cfg,scale", [(2, 0.5), (3, 0.25), (3, 0.5)])
def test_block_owners_in_the_shadow_pass_keep_their_map(pkg, hip, cfg, scale):
    """owners alone STORE their blocks (the store is the clear); beside the small path the map is cleared, the set-up kernel draws the small triangles and the
    owners of non-empty bins merge into it: the same map either way, and with the integer rasteriser forced (classic owners, no small path)"""
    sc = pkg.scenes.CONFIGS[cfg](scale=scale)
    ref, s0 = shadow(hip, sc, 0)
    for small, debug in ((1, 0), (0, 0), (1, 32)):
        m, st = shadow(hip, sc, small, owner=3, debug=debug)
        np.testing.assert_array_equal(m, ref)
        assert st[2] == s0[2]


def test_whole_frames_with_a_moving_sun(pkg, hip):
    """frames in flight redraw the map beside the previous frame's shading: same frames with and without the small path"""
    sc = pkg.scenes.config3(scale=0.25)
    frames = []
    for small in (0, 1):
        r = sc.upload(hip.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
        r.set_option("small_triangles", small); r.set_option("shadow_cache", 0)
        out = []
        import copy
        for k in range(4):
            d = copy.deepcopy(sc.desc)
            d.sun["rotation"] = (-70.0 + 3.0 * k, 12.0 + 5.0 * k)
            out.append(r.render_frame(d, sc.settings).copy())
        frames.append(out)
        r.close()
    for a, b in zip(*frames):
        np.testing.assert_array_equal(a, b)
    assert not np.array_equal(frames[0][0], frames[0][3])


def test_option_values(pkg, hip):
    sc = pkg.scenes.config2(scale=0.1)
    r = sc.upload(hip.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
    for bad in (-1, 2):
        with pytest.raises(hip.ArcticError):
            r.set_option("small_triangles", bad)
    r.close()
