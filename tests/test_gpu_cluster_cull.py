"""ARCTIC_OPT_CLUSTER_CULL (round 5): the prepasses skip workgroups of 256 triangles / 256 vertices whose object-space box lies beyond a side
of the pass's scissor rectangle, the near or the far plane (needs an MI355X).

The reference draws every object and leaves the rest to the hardware's clipper and culler (src/renderer/forward_pass.cpp:212-224,
src/renderer/shadow_map_pass.cpp:157-167): there is one answer per pixel, so skipping must change nothing -- visibility plane, interpolated
attributes, shadow map, record and work-item counts, the frame shaded from the visibility plane (which reads the transformed vertices of every
surviving record: a vertex block skipped wrongly shows there) -- bit for bit, for cameras inside and outside the scene, shards, objects that
straddle the scissor's sides by fractions of a pixel, and meshes the host cannot bound.
"""
import copy

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

COUNT = 1024     # ARCTIC_OPT_DEBUG bit 10: count what was skipped


def prepass(hip, sc, cull, desc=None, frame=True, **kw):
    desc = desc or sc.desc
    r = sc.upload(hip.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights, **kw))
    r.set_option("cluster_cull", cull)
    r.set_option("debug", COUNT)
    if sc.shadow_size:
        r.pass_shadow_map(desc)
    r.pass_gbuffer(desc)
    counts = [r.cull_counts(False).copy(), r.cull_counts(True).copy() if sc.shadow_size else np.zeros(4, np.uint32)]
    attrs, mat, depth, tri = r.read_gbuffer()
    out = [attrs.view(np.uint32).copy(), mat.copy(), depth.view(np.uint32).copy(), tri.copy(),
           r.read_shadow_map().view(np.uint32).copy() if sc.shadow_size else np.zeros(1, np.uint32), r.stats()[:4].copy()]
    if frame:
        out.append(r.render_frame(desc, sc.settings).copy())
        out.append(r.render_frame(desc, sc.settings).copy())     # (the second one with two frames in flight's other set of tables)
    r.close()
    return out, counts


def same(a, b):
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("cfg,scale", [(1, 0.5), (2, 0.25), (3, 0.2), (3, 0.5)])
def test_culling_changes_nothing(pkg, hip, cfg, scale):
    sc = pkg.scenes.CONFIGS[cfg](scale=scale)
    (off, c0), (setup_only, c1), (both, c3) = (prepass(hip, sc, m) for m in (0, 1, 3))
    assert (off[3] != 0xFFFFFFFF).mean() > 0.1
    same(off, setup_only)
    same(off, both)
    assert c0[0][1] == 0 and c0[0][3] == 0 and c1[0][3] == 0
    assert c1[0][1] == c3[0][1]
    print(f"config {cfg} x{scale}: forward pass skips {c3[0][1]} of {c3[0][0]} clusters and {c3[0][3]} of {c3[0][2]} vertex blocks, shadow pass {c3[1][1]} of {c3[1][0]} / {c3[1][3]} of {c3[1][2]}")
    if cfg == 3 and scale >= 0.5:      # not vacuous: the camera stands inside the atrium, most of it is behind it or beside the view
        assert c3[0][1] > 0.3 * c3[0][0] and c3[0][3] > 0.2 * c3[0][2]


def cameras(rng, n):
    out = []
    for i in range(n):
        kind = i % 4
        if kind == 0:    # inside the hall, any direction
            eye = rng.uniform((-12, 0.5, -5), (12, 9, 5)); rot = (rng.uniform(-89, 89), rng.uniform(-180, 180))
        elif kind == 1:  # far outside, looking roughly at it (everything small, far plane close to mattering) or away from it (everything skipped)
            d = rng.normal(size=3); d /= np.linalg.norm(d); eye = d * rng.uniform(30, 400) + (0, 5, 0)
            yaw = np.degrees(np.arctan2(-d[0], -d[2])) + (180 if i % 8 == 1 else 0); rot = (rng.uniform(-30, 30), yaw)
        elif kind == 2:  # nose on a wall / the floor: the near plane cuts clusters, w <= 0 corners
            eye = (rng.uniform(-14, 14), rng.uniform(0.01, 0.3), rng.uniform(-6, 6)); rot = (rng.uniform(-60, 10), rng.uniform(-180, 180))
        else:            # straight up / down (degenerate yaw)
            eye = rng.uniform((-5, 1, -3), (5, 8, 3)); rot = (89.9 if i % 8 == 3 else -89.9, rng.uniform(-180, 180))
        out.append((tuple(float(x) for x in eye), tuple(float(x) for x in rot)))
    return out


def test_random_cameras(pkg, hip):
    sc = pkg.scenes.config3(scale=0.25)
    rng = np.random.default_rng(20251)
    skipped = []
    for k, (eye, rot) in enumerate(cameras(rng, 16)):
        desc = copy.deepcopy(sc.desc)
        desc.camera["eye"], desc.camera["rotation"] = eye, rot
        if k % 5 == 4:
            desc.camera["z_near_far"] = (0.5, 12.0)          # a far plane that cuts the hall
        if k % 3 == 2:
            desc.sun["rotation"] = (float(rng.uniform(-80, -20)), float(rng.uniform(-180, 180)))
        (a, _), (b, c) = prepass(hip, sc, 0, desc, frame=(k % 2 == 0)), prepass(hip, sc, 3, desc, frame=(k % 2 == 0))
        same(a, b)
        skipped.append(c[0][1] / c[0][0])
    print("fraction of the clusters skipped per camera:", " ".join(f"{x:.2f}" for x in skipped))
    assert max(skipped) > 0.5 and min(skipped) < max(skipped)


@pytest.mark.parametrize("kw", [dict(row_begin=0, row_end=48), dict(row_begin=96, row_end=160), dict(row_begin=263, row_end=264),
                                dict(band_rows=16, shard=(1, 3)), dict(band_rows=64, shard=(2, 4))], ids=lambda k: "-".join(f"{a}{b}" for a, b in k.items()))
def test_shards(pkg, hip, kw):
    """a row-range shard's scissor is its rows: clusters above and below it go too; interleaved bands keep the whole frame's rectangle"""
    sc = pkg.scenes.config3(scale=0.25)
    (a, _), (b, c) = prepass(hip, sc, 0, **kw), prepass(hip, sc, 3, **kw)
    same(a, b)
    (_, whole) = prepass(hip, sc, 3, frame=False)
    if "row_begin" in kw:
        assert c[0][1] >= whole[0][1]
    else:
        assert c[0][1] == whole[0][1]


def test_objects_on_the_scissors_sides(pkg, hip):
    """one small dense quad (several clusters) moved across every side of the frame in steps of a fraction of a pixel, and through the near and the
    far plane: whatever is skipped, the image's bits are those of the unculled prepass"""
    S = pkg.scenes
    rng = np.random.default_rng(7)
    mats = [S.make_material_textures(rng, 64)]
    meshes = [S.quad((-0.05, -0.05, 0.0), (0.1, 0, 0), (0, 0.1, 0), 24, 24) + (0,)]      # 1152 triangles: 5 clusters, strips of the quad
    w, h = 256, 144
    cam = dict(eye=(0.0, 0.0, 3.0), rotation=(0.0, -90.0), aspect=w / h, fov_y=45.0, z_near_far=(0.1, 50.0))      # looks down -z
    # where does the quad's plane meet the frustum's sides?  half-extent of the view at the quad's distance
    half_h = 3.0 * np.tan(np.radians(22.5)); half_w = half_h * w / h
    px = 2 * half_w / w
    offsets = []
    for side, centre in (("x", half_w), ("x", -half_w), ("y", half_h), ("y", -half_h)):
        for d in np.linspace(-2.5, 2.5, 11):
            t = centre + np.sign(centre) * (0.05 + d * px)      # the quad's near edge d pixels inside (negative) / outside the side
            offsets.append((t, 0.0, 0.0) if side == "x" else (0.0, t, 0.0))
    for z in (3.0 - 0.1 + 1e-4, 3.0 - 0.1 - 1e-4, 3.0 - 0.1, 3.0 - 50.0 + 1e-3, 3.0 - 50.0 - 1e-3, 3.0 + 5.0):      # at the near plane, the far plane, behind the camera
        offsets.append((0.0, 0.0, z))
    hits = seen = 0
    ra, rb = [hip.Renderer(w, h, 0, 16) for _ in range(2)]
    for r, cull in ((ra, 0), (rb, 3)):
        r.create_material(*mats[0]); r.create_mesh(*meshes[0][:2], 0)
        r.set_option("cluster_cull", cull); r.set_option("debug", COUNT)
    for off in offsets:
        desc = S.SceneDesc(camera=cam, ambient=0.1, sun=S.DEFAULT_SUN, objects=S.make_objects([(S.translation(*off), 0)]))
        out = []
        for r in (ra, rb):
            r.pass_gbuffer(desc)
            attrs, mat, depth, tri = r.read_gbuffer()
            out.append([attrs.view(np.uint32).copy(), mat.copy(), depth.view(np.uint32).copy(), tri.copy(), r.stats()[:2].copy()])
        same(out[0], out[1])
        n = rb.cull_counts(False)
        drawn = (out[0][3] != 0xFFFFFFFF).any()
        assert not (drawn and n[1] == n[0])
        seen += int(drawn)
        hits += int(drawn and n[1] > 0)      # partly in, and some of its strips skipped
    assert seen >= 12 and hits >= 4, (seen, hits)
    ra.close(); rb.close()


def test_meshes_the_host_cannot_bound(pkg, hip):
    """indices out of range draw nothing and bound nothing; a position that is not finite makes its blocks' boxes infinite (never skipped): the
    prepass is the unculled one's, whatever that draws"""
    S = pkg.scenes
    rng = np.random.default_rng(9)
    mats = [S.make_material_textures(rng, 64)]
    v, i = S.quad((-1, -1, 0), (2, 0, 0), (0, 2, 0), 40, 40)
    v = v.copy(); i = i.copy()
    i[5 * 3] = len(v) + 7                    # one triangle with an index out of range
    i[300 * 3 + 1] = 0xFFFFFFFF
    v["position"][len(v) // 2] = (np.nan, 0.0, np.inf)
    cam = dict(eye=(0.0, 0.0, 3.0), rotation=(0.0, -90.0), aspect=2.0, fov_y=45.0, z_near_far=(0.1, 50.0))
    out = []
    for cull in (0, 3):
        r = hip.Renderer(256, 128, 0, 16)
        r.create_material(*mats[0]); r.create_mesh(v, i, 0)
        r.set_option("cluster_cull", cull)
        frames = []
        for off in ((0, 0, 0), (4.0, 0, 0), (0, 0, 10.0)):
            desc = S.SceneDesc(camera=cam, ambient=0.1, sun=S.DEFAULT_SUN, objects=S.make_objects([(S.translation(*off), 0)]))
            r.pass_gbuffer(desc)
            _, mat, depth, tri = r.read_gbuffer()
            frames += [mat.copy(), depth.view(np.uint32).copy(), tri.copy()]
        out.append(frames)
        r.close()
    same(out[0], out[1])


def test_option_is_validated(pkg, hip):
    r = hip.Renderer(64, 64, 0, 16)
    for bad in (2, 4, -1):
        with pytest.raises(hip.ArcticError):
            r.set_option("cluster_cull", bad)
    with pytest.raises(hip.ArcticError):
        r.cull_counts(False)                # nothing counted yet
    r.close()
