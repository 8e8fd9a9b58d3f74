"""Block ownership in the prepasses' rasteriser (ARCTIC_OPT_RASTER_OWNER, round 3): work items binned per 16x16 block, one
owner wave per block writing it once, the merging atomicMin rasteriser only for what the bins do not hold.  The D3D12 raster it
stands in for (forward_pass.cpp:137-151,212-224, shadow_map_pass.cpp:96-97,157-167) has one answer per pixel, so the two
rasterisers must agree bit for bit -- depth, winning triangle, shadow map -- on whole targets, ragged sizes, row-range and
interleaved shards, bins that overflow, and records that take the integer path.  (The library's own choice is owners only for
handles that own 4 Mpx of the frame or more: against the ORACLE they run in the full-size tests -- 4K, 8K -- and in
test_gpu_parity.py's `large-frame defaults` case, which forces them, with two tiles per wave, on a small frame.)"""
import copy

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def prepass(hip, sc, owner, size=None, shadow=None, debug=0, **kw):
    w, h = size or (sc.width, sc.height)
    r = sc.upload(hip.Renderer(w, h, sc.shadow_size if shadow is None else shadow, sc.max_lights, **kw))
    r.set_option("raster_owner", 3 * owner)      # both passes with owners, or neither
    r.set_option("small_triangles", 0)           # same() also compares the WORK-ITEM counts of the two rasterisers: one producer for both (k_setup draws small shadow triangles
                                                 # itself only in front of the atomic rasteriser; that path and its owner case: tests/test_gpu_small_triangles.py)
    if debug:
        r.set_option("debug", debug)
    r.pass_shadow_map(sc.desc)
    r.pass_gbuffer(sc.desc)
    _, mat, depth, tri = r.read_gbuffer()
    out = (depth.view(np.uint32).copy(), tri.copy(), mat.copy(), r.read_shadow_map().view(np.uint32).copy(), r.stats().copy())
    r.close()
    return out


def same(a, b, work_too=True):
    for x, y in zip(a[:4], b[:4]):
        np.testing.assert_array_equal(x, y)
    if work_too:
        np.testing.assert_array_equal(a[4][:4], b[4][:4])      # records and work items: the same set-up whoever draws it


@pytest.mark.parametrize("cfg,scale", [(2, 0.25), (3, 0.2), (3, 0.07)])
def test_owner_raster_equals_atomic_raster(pkg, hip, cfg, scale):
    sc = pkg.scenes.CONFIGS[cfg](scale=scale)
    a, b = prepass(hip, sc, 1), prepass(hip, sc, 0)
    assert (a[1] != 0xFFFFFFFF).mean() > 0.1
    same(a, b)


@pytest.mark.parametrize("size,shadow", [((1, 1), 64), ((17, 9), 67), ((250, 131), 333), ((264, 152), 100)])
def test_ragged_targets(pkg, hip, size, shadow):
    """targets that end inside a block, inside a tile, or are smaller than one; shadow maps likewise"""
    sc = pkg.scenes.config3(scale=0.1)
    same(prepass(hip, sc, 1, size, shadow), prepass(hip, sc, 0, size, shadow))


@pytest.mark.parametrize("rows", [(0, 40), (37, 83), (8, 24), (112, 144), (143, 144)])
def test_row_range_shards(pkg, hip, rows):
    """row ranges that start / end inside a block and inside a tile: the owner writes only what the shard stores"""
    sc = pkg.scenes.config3(scale=0.07)
    assert sc.height == 144
    a = prepass(hip, sc, 1, row_begin=rows[0], row_end=rows[1])
    b = prepass(hip, sc, 0, row_begin=rows[0], row_end=rows[1])
    same(a, b)
    full = prepass(hip, sc, 1)
    for k in range(3):
        np.testing.assert_array_equal(a[k], full[k][rows[0]:rows[1]])


@pytest.mark.parametrize("band,world", [(16, 2), (16, 3), (32, 5), (8, 2), (24, 3)])
def test_interleaved_shards(pkg, hip, band, world):
    """bands of an even number of tile rows (the owner grid walks the shard's own block rows) and of an odd number (a block's
    two tile rows belong to different shards)"""
    from arctic_renderer_amd.sharding import owned_rows
    sc = pkg.scenes.config3(scale=0.07)
    full = prepass(hip, sc, 1)
    for rank in range(world):
        a = prepass(hip, sc, 1, band_rows=band, shard=(rank, world))
        b = prepass(hip, sc, 0, band_rows=band, shard=(rank, world))
        same(a, b)
        rows = owned_rows(sc.height, rank, world, band)
        for k in range(3):
            np.testing.assert_array_equal(a[k], full[k][rows])


def test_full_bins_fall_through_to_the_atomic_raster(pkg, oracle, hip):
    """a finely tessellated sphere on a small target: hundreds of work items per block, 32 bin slots -- most of the frame is
    drawn by the atomic rasteriser on top of what the owners wrote.  Checked against the oracle too."""
    rng = np.random.default_rng(5)
    S = pkg.scene
    v, i = pkg.scenes.uv_sphere(1.2, 192, 96)
    mats = [pkg.scenes.make_material_textures(rng, 32)]
    lights = pkg.scenes.random_lights(rng, 2, (-2, -2, 0.5), (2, 2, 3))
    desc = S.SceneDesc(camera=dict(eye=(0, 0, 4.0), rotation=(0.0, -90.0), aspect=96 / 64, fov_y=60.0, z_near_far=(0.1, 100.0)),
                       ambient=0.1, sun=dict(position=(2, 10, 6), rotation=(-55.0, -110.0), color=(8, 8, 8)),
                       objects=S.make_objects([(np.eye(4), 0)]), point_lights=lights)
    outs = []
    for cls, owner in ((oracle.Oracle, None), (hip.Renderer, 1), (hip.Renderer, 0)):
        r = cls(96, 64, 128, 16)
        r.create_material(*mats[0]); r.create_mesh(v, i, 0); r.update_lights(lights)
        if owner is not None:
            r.set_option("raster_owner", 3 * owner)
            r.set_option("small_triangles", 0)   # (the sphere's shadow triangles are all small: drawn by the set-up kernel they would leave the shadow pass's bins empty)
        r.pass_shadow_map(desc); r.pass_gbuffer(desc)
        g = r.read_gbuffer()
        outs.append([x.view(np.uint32) for x in g] + [r.read_shadow_map().view(np.uint32)])
        if owner == 1:
            st = r.stats()
            assert int(st[1]) > 2 * 24 * 32      # more than twice the work items all 24 bins of the target hold ...
            assert int(st[10]) >= int(st[1]) - 24 * 32 and int(st[11]) > 0      # ... so most of them were left to the atomic rasteriser
            counts = r.bin_counts()
            assert counts.shape == (4, 6) and counts.max() >= 32
        r.close()
    for other in outs[1:]:
        for x, y in zip(outs[0], other):
            np.testing.assert_array_equal(x, y)


def test_integer_records_are_left_to_the_atomic_raster(pkg, hip):
    """ARCTIC_OPT_DEBUG bit 5 sends every record down the 64-bit integer path: no item may be binned, the owners only clear"""
    sc = pkg.scenes.config3(scale=0.07)
    same(prepass(hip, sc, 1, debug=32), prepass(hip, sc, 1), work_too=False)   # (integer records take every block of their bounds)
    same(prepass(hip, sc, 1, debug=32), prepass(hip, sc, 0, debug=32))


def test_frames_keep_their_bytes(pkg, hip):
    """whole frames (two in flight, shadow redraw beside the prepass) with and without owners"""
    sc = pkg.scenes.config3(scale=0.1)
    imgs = []
    for owner in (1, 0):
        r = sc.upload(hip.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
        r.set_option("raster_owner", 3 * owner)
        frames = []
        for k in range(5):
            d = copy.deepcopy(sc.desc)
            d.camera["rotation"] = (-15.0 + k, 7.0 * k)
            if k >= 2:
                d.sun = dict(d.sun, rotation=(d.sun["rotation"][0] - 1.5 * k, d.sun["rotation"][1] + 4.0 * k))
            frames.append(r.render_frame(d, sc.settings).copy())
        imgs.append(frames)
        r.close()
    for a, b in zip(*imgs):
        np.testing.assert_array_equal(a, b)
