"""Parity of the HIP path against the CPU oracle, through the C-ABI (needs an MI355X).

Bars (BASELINE.json north_star):
  * shadow map, visibility, G-buffer: BIT-EXACT (integer coverage + identically ordered IEEE fp32)
  * shaded output: |HIP - oracle| <= 1e-4 per channel on the float LDR image (after tonemap + gamma,
    before the UNORM8 store); RGBA8 differs by at most 1 LSB on a small fraction of channels.
    The arbiter is the oracle's float64 evaluation of the HLSL formulas on the same fp32 inputs
    (oracle/arctic_oracle.cpp, "BRDF + tonemap, templated on the real type"): the literal fp32
    evaluation order of forward.hlsl:137 is itself only accurate to ~1e-3 on low-roughness
    highlights (tests/test_oracle_noise_floor.py measures that), so it cannot arbitrate 1e-4.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-4   # north_star: "<= 1e-4 per-channel deviation"


def build_pair(pkg, oracle, hip, sc, **kw):
    o = sc.upload(oracle.Oracle(sc.width, sc.height, sc.shadow_size, sc.max_lights, **kw))
    r = sc.upload(hip.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights, **kw))
    r.set_option("keep_float_output", 1)
    return o, r


# configs 4/5: 256 and 1024 point lights.  The last case runs the paths the library chooses by itself only for LARGE frames -- block owners in the
# forward raster (from 4 Mpx), two tiles per wave (from ~3 Mpx) -- on a small one, so that they meet the oracle directly in this suite too
SCENES = [(1, 0.5, {}), (2, 0.25, {}), (3, 0.1, {}), (3, 0.2, {}), (4, 0.06, {}), (5, 0.03, {}), (3, 0.15, {"raster_owner": 1, "tiles_per_wave": 2})]


@pytest.fixture(scope="module", params=SCENES, ids=[f"config{c}-x{s}" + ("-large-frame-defaults" if o else "") for c, s, o in SCENES])
def pair(request, pkg, oracle, hip):
    cfg, scale, options = request.param
    sc = pkg.scenes.CONFIGS[cfg](scale=scale)
    o, r = build_pair(pkg, oracle, hip, sc)
    for name, value in options.items():
        r.set_option(name, value)
    o.pass_shadow_map(sc.desc)
    o.pass_gbuffer(sc.desc)
    r.pass_shadow_map(sc.desc)
    r.pass_gbuffer(sc.desc)
    yield sc, o, r
    r.close()
    o.close()


def test_shadow_map_bit_exact(pair):
    sc, o, r = pair
    if not sc.shadow_size:
        pytest.skip("config has no shadow map")
    a, b = o.read_shadow_map(), r.read_shadow_map()
    assert (a < 1.0).any(), "shadow map is empty: the test would be vacuous"
    np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))


def test_gbuffer_bit_exact(pair):
    sc, o, r = pair
    oa, om, od, ot = o.read_gbuffer()
    ha, hm, hd, ht = r.read_gbuffer()
    assert (om != 0xFFFFFFFF).mean() > 0.1
    np.testing.assert_array_equal(om, hm)
    np.testing.assert_array_equal(ot, ht)
    np.testing.assert_array_equal(od.view(np.uint32), hd.view(np.uint32))
    np.testing.assert_array_equal(oa.view(np.uint32), ha.view(np.uint32))
    s_o, s_h = o.stats(), r.stats()
    assert s_o[0] == s_h[0] and s_o[2] == s_h[2]   # same number of set-up triangles (forward, shadow)


def test_integer_rasteriser_equals_binary64_rasteriser(pair):
    """debug bit 5 sends every record through the 64-bit integer rasteriser (the path of records with coordinates of 2^24 and
    more) instead of the binary64 planes: same shadow map, same visibility, same G-buffer, bit for bit."""
    sc, o, r = pair
    ref_map = r.read_shadow_map().copy() if sc.shadow_size else None
    ref_g = [x.copy() for x in r.read_gbuffer()]
    r.set_option("debug", 32)
    try:
        r.pass_shadow_map(sc.desc); r.pass_gbuffer(sc.desc)
        if sc.shadow_size:
            np.testing.assert_array_equal(r.read_shadow_map().view(np.uint32), ref_map.view(np.uint32))
        for a, b in zip(ref_g, r.read_gbuffer()):
            np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))
    finally:
        r.set_option("debug", 0)
        r.pass_shadow_map(sc.desc); r.pass_gbuffer(sc.desc)


@pytest.mark.parametrize("tm", [0, 1, 2], ids=["reinhard", "exposure", "aces"])
def test_shade_parity(pair, tm):
    sc, o, r = pair
    settings = (tm, 2.2, 1.0 if tm != 1 else 0.7)
    o.pass_shade(sc.desc, settings)
    r.pass_shade(sc.desc, settings)
    oldr, ohdr, orgba = o.read_output()
    hldr, hhdr, hrgba = r.read_output()
    assert np.isfinite(hldr).all()
    err = np.abs(hldr - oldr)
    assert err.max() <= TOL, f"max err {err.max():.3e} at {np.unravel_index(err.argmax(), err.shape)}"
    d = np.abs(hrgba.astype(np.int16) - orgba.astype(np.int16))
    assert d.max() <= 1
    assert (d != 0).mean() < 2e-3
    assert (hrgba[..., 3] == 255).all()


def test_render_frame_end_to_end(pair):
    sc, o, r = pair
    ref = o.render_frame(sc.desc, sc.settings)
    img = r.render_frame(sc.desc, sc.settings)
    d = np.abs(img.astype(np.int16) - ref.astype(np.int16))
    assert d.max() <= 1 and (d != 0).mean() < 2e-3


def test_culling_is_exact(pair):
    """the wave-level culling only skips work that multiplies to exactly zero."""
    sc, o, r = pair
    r.set_option("culling", 1)
    r.pass_shade(sc.desc, sc.settings)
    a = r.read_output()[0].copy()
    r.set_option("culling", 0)
    r.pass_shade(sc.desc, sc.settings)
    b = r.read_output()[0]
    r.set_option("culling", 1)
    assert np.abs(a - b).max() <= 2e-6


def test_frame_from_visibility_plane_equals_frame_through_gbuffer(pair):
    """ARCTIC_OPT_VISBUFFER (default): arctic_render_frame interpolates the attributes inside the shading kernel instead of
    writing and re-reading the G-buffer.  Same operations in the same order: the float image is bit-identical, and the
    G-buffer materialised afterwards on request is the prepass's G-buffer."""
    sc, o, r = pair
    r.set_option("visbuffer", 0)
    a = r.render_frame(sc.desc, sc.settings)
    a_ldr, a_hdr, _ = (x.copy() for x in r.read_output())
    ga = r.read_gbuffer()
    r.set_option("visbuffer", 1)
    b = r.render_frame(sc.desc, sc.settings)
    b_ldr, b_hdr, _ = r.read_output()
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(a_ldr.view(np.uint32), b_ldr.view(np.uint32))
    np.testing.assert_array_equal(a_hdr.view(np.uint32), b_hdr.view(np.uint32))
    gb = r.read_gbuffer()                       # resolved now, from the visibility plane of that frame
    for x, y in zip(ga, gb):
        np.testing.assert_array_equal(x.view(np.uint32), y.view(np.uint32))
    r.pass_shade(sc.desc, sc.settings)          # and the G-buffer API keeps working after such a frame
    np.testing.assert_array_equal(r.read_output()[0].view(np.uint32), a_ldr.view(np.uint32))
    r.set_option("debug", 64)                   # the same frame with 64-bit pointer gathers (the path of tables of 4 GiB and more)
    try:
        c = r.render_frame(sc.desc, sc.settings)
        np.testing.assert_array_equal(c, b)
        np.testing.assert_array_equal(r.read_output()[0].view(np.uint32), a_ldr.view(np.uint32))
    finally:
        r.set_option("debug", 0)


def test_fast_tile_equals_general_tile(pair):
    """shade.hip has two code paths per 8x8 tile: the fast tile (whole tile inside the target, one packed material, shadow test
    decided by the bounds table: straight-line code) and the general tile (everything else).  ARCTIC_OPT_DEBUG bit 8 sends every
    tile through the general one.  Both call the same arithmetic in the same order: float planes and RGBA8 are bit-identical,
    through the G-buffer and through the visibility plane, with either light loop, with and without exact culling."""
    sc, o, r = pair
    try:
        for path in (1, 2):
            for culling in (1, 0):
                r.set_option("light_path", path); r.set_option("culling", culling)
                outs = []
                for dbg in (0, 256):
                    r.set_option("debug", dbg)
                    r.pass_shadow_map(sc.desc); r.pass_gbuffer(sc.desc)
                    r.pass_shade(sc.desc, sc.settings)
                    ldr, hdr, rgba = (x.copy() for x in r.read_output())
                    frame = r.render_frame(sc.desc, sc.settings).copy()
                    outs.append((ldr, hdr, rgba, frame, r.read_output()[0].copy()))
                for a, b in zip(*outs):
                    np.testing.assert_array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b)
    finally:
        r.set_option("debug", 0); r.set_option("light_path", 0); r.set_option("culling", 1)
        r.pass_shadow_map(sc.desc); r.pass_gbuffer(sc.desc)


def test_light_paths_agree(pair, pkg):
    """ARCTIC_OPT_LIGHT_PATH: the shading kernel runs the light loop scalar (1: lights through the scalar cache) or two
    lights at a time in packed fp32 (2), both through the scalar cache; 0 picks by light count.  Same formulas: the float images agree to fp32
    rounding (the compiler contracts differently per kernel), each is within the parity bar of the oracle -- with the
    scene's own lights and as a sun-only scene, through the G-buffer and through the visibility-plane frame."""
    sc, o, r = pair
    lights = sc.lights
    try:
        for subset in (lights, lights[:1], lights[:0]):   # an odd count exercises the black pad light of the last pair
            r.update_lights(subset); o.update_lights(subset)
            o.pass_shade(sc.desc, sc.settings)
            ref = o.read_output()[0]
            outs = {}
            for path in (1, 2, 0):
                r.set_option("light_path", path)
                r.pass_shadow_map(sc.desc); r.pass_gbuffer(sc.desc)
                r.pass_shade(sc.desc, sc.settings)
                ldr, hdr, rgba = (x.copy() for x in r.read_output())
                frame = r.render_frame(sc.desc, sc.settings)
                outs[path] = (ldr, hdr, rgba, frame, r.read_output()[0].copy())
                assert np.abs(ldr - ref).max() <= TOL and np.abs(outs[path][4] - ref).max() <= TOL
            for path in (2, 0):
                for k in (0, 4):
                    assert np.abs(outs[1][k] - outs[path][k]).max() <= 3e-6
                assert np.abs(outs[1][1] - outs[path][1]).max() <= 3e-6 * max(1.0, float(outs[1][1].max()))
                for k in (2, 3):
                    assert np.abs(outs[1][k].astype(np.int16) - outs[path][k].astype(np.int16)).max() <= 1
    finally:
        r.set_option("light_path", 0)
        r.update_lights(lights); o.update_lights(lights)
        r.pass_shadow_map(sc.desc); r.pass_gbuffer(sc.desc)


def test_shadow_bounds_table_changes_nothing(pair):
    """the min/max table of the shadow map (k_shadow_bounds) only decides pixels whose 25 PCF compares all agree; debug
    bit 3 runs the shadow test without it, bit 4 runs the 25-tap path of edge tiles from a per-wave LDS tile instead of the
    per-lane register window: byte-identical float images in every combination.  Also after the map was replaced from the host."""
    sc, o, r = pair
    r.pass_shade(sc.desc, sc.settings)
    a = [x.copy() for x in r.read_output()]
    try:
        for bits in (8, 16, 24):
            r.set_option("debug", bits)
            r.pass_shade(sc.desc, sc.settings)
            b = r.read_output()
            for x, y in zip(a, b):
                np.testing.assert_array_equal(x.view(np.uint32) if x.dtype == np.float32 else x, y.view(np.uint32) if y.dtype == np.float32 else y)
            frame = r.render_frame(sc.desc, sc.settings)          # the visibility-plane kernels have the same variants
            np.testing.assert_array_equal(frame, a[2])
            r.pass_gbuffer(sc.desc)
        r.set_option("debug", 0)
        if sc.shadow_size:
            m = r.read_shadow_map()
            m2 = np.ascontiguousarray(m[::-1, ::-1])
            r.write_shadow_map(m2); o.write_shadow_map(m2)
            r.pass_shade(sc.desc, sc.settings); o.pass_shade(sc.desc, sc.settings)
            assert np.abs(r.read_output()[0] - o.read_output()[0]).max() <= TOL
            c = [x.copy() for x in r.read_output()]
            r.set_option("debug", 8)
            r.pass_shade(sc.desc, sc.settings)
            np.testing.assert_array_equal(c[0].view(np.uint32), r.read_output()[0].view(np.uint32))
            r.write_shadow_map(m); o.write_shadow_map(m)
    finally:
        r.set_option("debug", 0)
        r.pass_shadow_map(sc.desc); r.pass_gbuffer(sc.desc)


def test_light_statistics(pair):
    """ARCTIC_OPT_COUNT_LIGHT_EVALS: the counting variant of the kernel renders the same image and its counters are
    consistent: lit pixels = pixels the oracle gives 1 - shadow != 0, evaluations = lit pixels x lights, contributing <= evaluations."""
    sc, o, r = pair
    r.pass_shade(sc.desc, sc.settings)
    a = r.read_output()[0].copy()
    try:
        for path in (1, 2):
            r.set_option("light_path", path)
            r.set_option("count_light_evals", 1)
            r.pass_shade(sc.desc, sc.settings)
            st = r.stats()
            assert np.abs(r.read_output()[0] - a).max() <= 3e-6
            assert int(st[5]) == int(st[6]) * len(sc.lights)
            assert int(st[7]) <= int(st[5])
            assert int(st[9]) * 64 >= int(st[6]) and int(st[8]) <= int(st[9]) * max(len(sc.lights), 1)
            r.set_option("count_light_evals", 0)
    finally:
        r.set_option("count_light_evals", 0)
        r.set_option("light_path", 0)


# ---------------------------------------------------------------------------------------------------------------
# beyond the per-scene fixtures: sharding, odd inputs, the stand-alone post-process kernel, full-size properties
# ---------------------------------------------------------------------------------------------------------------
def test_row_shards_equal_full_frame(pkg, oracle, hip):
    """multi-GPU invariant: the frame assembled from row shards is bit-identical to the single-device frame.
    The cut at row 37 is deliberately not a multiple of the 8-pixel tile."""
    sc = pkg.scenes.config3(scale=0.1)
    full = sc.upload(hip.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
    ref = full.render_frame(sc.desc, sc.settings)
    for cuts in ([0, 37, sc.height], [0, 64, 128, sc.height]):
        parts = []
        for b, e in zip(cuts, cuts[1:]):
            r = sc.upload(hip.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights, row_begin=b, row_end=e))
            parts.append(r.render_frame(sc.desc, sc.settings))
            assert parts[-1].shape == (e - b, sc.width, 4)
            r.close()
        np.testing.assert_array_equal(np.concatenate(parts, 0), ref)
    full.close()


def test_interleaved_band_shards_equal_full_frame(pkg, hip):
    """the load-balanced sharding the multi-GPU bench uses: bands of 16 rows dealt round-robin over the ranks."""
    from importlib import import_module
    sh = import_module("arctic_renderer_amd.sharding")
    import torch
    sc = pkg.scenes.config3(scale=0.1)                      # 384 x 216: the last band is partial (216 = 13.5 x 16)
    full = sc.upload(hip.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
    ref = full.render_frame(sc.desc, sc.settings)
    full.close()
    for world, band in ((3, 16), (8, 16), (2, 8)):
        parts = []
        for rank in range(world):
            r = sc.upload(hip.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights, band_rows=band, shard=(rank, world)))
            rows = sh.owned_rows(sc.height, rank, world, band)
            assert r.rows == len(rows)
            img = r.render_frame(sc.desc, sc.settings)
            np.testing.assert_array_equal(img, ref[rows])
            parts.append(torch.from_numpy(img))
            r.close()
        np.testing.assert_array_equal(sh.assemble_banded(parts, sc.height, band).numpy(), ref)


@pytest.mark.parametrize("cfg,scale", [(1, 0.5), (2, 0.25), (3, 0.1)])
def test_skybox_parity(pkg, oracle, hip, cfg, scale):
    """pixels without geometry take the environment map along their view ray (skybox.hlsl:61-90).
    Same 1e-4 bar; the geometry pixels must not change; shards see the sky of their own rows."""
    from importlib import import_module
    sh = import_module("arctic_renderer_amd.sharding")
    sc = pkg.scenes.CONFIGS[cfg](scale=scale)
    o, r = build_pair(pkg, oracle, hip, sc)
    o.pass_shadow_map(sc.desc); o.pass_gbuffer(sc.desc)
    r.pass_shadow_map(sc.desc); r.pass_gbuffer(sc.desc)
    r.pass_shade(sc.desc, sc.settings)
    before = r.read_output()[0].copy()
    env = pkg.scenes.synthetic_hdri(512, 256)
    o.create_hdri(env); r.create_hdri(env)
    covered = o.read_gbuffer()[1] != 0xFFFFFFFF
    for tm, hdr16 in ((0, 0), (2, 0), (1, 1)):
        o.set_hdr16(hdr16); r.set_option("hdr16", hdr16)
        settings = (tm, 2.2, 0.7)
        o.pass_shade(sc.desc, settings); r.pass_shade(sc.desc, settings)
        oldr, ohdr, _ = o.read_output()
        hldr, hhdr, _ = r.read_output()
        if (~covered).any():
            assert (hhdr[~covered].sum(-1) > 0).all(), "sky pixels stayed black"
            if not hdr16:
                assert np.abs(hhdr[~covered] - ohdr[~covered]).max() <= 1e-5 * max(1.0, ohdr[~covered].max())
        err = np.abs(hldr - oldr)
        if hdr16:   # the binary16 rounding is a discontinuity: same bar as test_reference_quantised_mode
            assert np.quantile(err, 0.999) <= TOL and err.max() <= 1e-3
        else:
            assert err.max() <= TOL, f"max err {err.max():.3e} at {np.unravel_index(err.argmax(), err.shape)}"
    o.set_hdr16(0); r.set_option("hdr16", 0)
    r.pass_shade(sc.desc, sc.settings)
    after = r.read_output()[0]
    np.testing.assert_array_equal(after[covered], before[covered])
    ref = r.render_frame(sc.desc, sc.settings)
    # a row shard with an unaligned cut and an interleaved band shard reproduce their rows of the frame
    cut = sc.height // 3 + 3
    rs = sc.upload(hip.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights, row_begin=cut, row_end=sc.height))
    rs.create_hdri(env)
    np.testing.assert_array_equal(rs.render_frame(sc.desc, sc.settings), ref[cut:])
    rs.close()
    rb = sc.upload(hip.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights, band_rows=16, shard=(1, 3)))
    rb.create_hdri(env)
    np.testing.assert_array_equal(rb.render_frame(sc.desc, sc.settings), ref[sh.owned_rows(sc.height, 1, 3, 16)])
    rb.close(); r.close(); o.close()


def test_render_frame_redraws_the_shadow_map_only_when_its_inputs_change(pkg, oracle, hip):
    """ARCTIC_OPT_SHADOW_CACHE (default on): same frames as with the cache off, through sun and object changes."""
    import copy
    sc = pkg.scenes.config2(scale=0.25)
    cached = sc.upload(hip.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
    plain = sc.upload(hip.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
    plain.set_option("shadow_cache", 0)
    serial = sc.upload(hip.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))   # debug bit 7: the shadow pass on the main stream, before
    serial.set_option("shadow_cache", 0); serial.set_option("debug", 128)                  # the visibility prepass instead of beside it
    desc = copy.deepcopy(sc.desc)
    frames = []
    for step in range(6):
        if step == 2:
            desc.sun["rotation"] = (-55.0, 30.0)                      # the light moves
        if step == 4:
            desc.objects["trs"][0][12] += 0.75                        # an object moves (glm column 3 = translation)
        if step == 5:
            desc.camera["eye"] = tuple(np.add(desc.camera["eye"], (0.2, 0.1, 0.0)))   # only the camera moves: map is reused
        a, b = cached.render_frame(desc, sc.settings), plain.render_frame(desc, sc.settings)
        np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(a, serial.render_frame(desc, sc.settings))
        np.testing.assert_array_equal(cached.read_shadow_map().view(np.uint32), plain.read_shadow_map().view(np.uint32))
        frames.append(a)
    assert (frames[0] == frames[1]).all() and (frames[1] != frames[2]).any() and (frames[3] != frames[4]).any()
    o = sc.upload(oracle.Oracle(sc.width, sc.height, sc.shadow_size, sc.max_lights))
    ref = o.render_frame(desc, sc.settings)
    d = np.abs(frames[-1].astype(np.int16) - ref.astype(np.int16))
    assert d.max() <= 1 and (d != 0).mean() < 2e-3
    cached.close(); plain.close(); serial.close(); o.close()


@pytest.mark.parametrize("in_flight", [2, 3])
def test_frames_in_flight_give_the_same_frames(pkg, oracle, hip, in_flight):
    """ARCTIC_OPT_FRAMES_IN_FLIGHT = 2 (default) or 3 (the reference's count, rhi.hpp:25: three sets, consecutive prepasses on two
    streams): the visibility prepass of a frame runs on its own stream, into the other set of
    tables (and, when the map is redrawn, the other shadow map), beside the shading of the frame before.  A run of frames whose camera
    moves every frame and whose objects and sun move now and then -- each frame enqueued without waiting for the one before -- must equal the same run with one frame at a time, byte for
    byte, also with pass-level calls (which see the latest frame's set) and G-buffer read-backs in between."""
    import copy
    import torch
    sc = pkg.scenes.config3(scale=0.25)
    two = sc.upload(hip.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
    two.set_option("frames_in_flight", in_flight)
    one = sc.upload(hip.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
    one.set_option("frames_in_flight", 1)
    n = 12
    outs = [[torch.empty((sc.height, sc.width, 4), dtype=torch.uint8, device="cuda") for _ in range(n)] for _ in range(2)]
    descs = []
    desc = copy.deepcopy(sc.desc)
    for k in range(n):
        desc = copy.deepcopy(desc)
        desc.camera["eye"] = (0.4 * k - 2.0, 5.0 + 0.1 * k, 0.3 * np.sin(k))
        desc.camera["rotation"] = (-15.0 + k, 7.0 * k)
        if k in (4, 9):
            desc.objects["trs"][8 + k][12] += 0.5          # a column moves: its table set is uploaded again, the other one a frame later
        if k in (2, 3, 6, 10):
            desc.sun = dict(desc.sun, rotation=(desc.sun["rotation"][0] - 1.5, desc.sun["rotation"][1] + 4.0))   # the sun moves: the map is redrawn, into the other shadow map
        descs.append(desc)
    for k in range(n):                                      # no flush inside the loop: frames are in flight
        for r, o in zip((two, one), outs):
            r.render_frame_device(descs[k], sc.settings, o[k].data_ptr())
        if k == 5:                                          # the lazily resolved G-buffer is the latest frame's
            ga, gb = two.read_gbuffer(), one.read_gbuffer()
            for x, y in zip(ga, gb):
                np.testing.assert_array_equal(x.view(np.uint32), y.view(np.uint32))
        if k == 7:                                          # pass-level calls between frames
            for r in (two, one):
                r.pass_gbuffer(descs[k]); r.pass_shade(descs[k], sc.settings)
            np.testing.assert_array_equal(two.read_output(want=("rgba8",))[2], one.read_output(want=("rgba8",))[2])
    two.flush(); one.flush()
    for k in range(n):
        np.testing.assert_array_equal(outs[0][k].cpu().numpy(), outs[1][k].cpu().numpy(), err_msg=f"frame {k}")
    assert any((outs[0][k] != outs[0][k + 1]).any().item() for k in range(n - 1))
    o = sc.upload(oracle.Oracle(sc.width, sc.height, sc.shadow_size, sc.max_lights))
    for k in (0, n - 1):
        ref = o.render_frame(descs[k], sc.settings)
        d = np.abs(outs[0][k].cpu().numpy().astype(np.int16) - ref.astype(np.int16))
        assert d.max() <= 1 and (d != 0).mean() < 2e-3, f"frame {k}"
    two.close(); one.close(); o.close()


def test_shadow_passes_on_two_streams_never_share_their_scratch(pkg, hip):
    """arctic_pass_shadow_map runs on the main stream, a frame in flight redraws the map on the handle's shadow stream -- and both use
    the ONE set of shadow-pass scratch (transformed vertices, records, work items, counters).  Pass-level shadow passes still queued on
    the main stream followed at once, without a flush, by a frame with another sun must give the frame one-frame-at-a-time gives."""
    import copy
    import torch
    sc = pkg.scenes.config3(scale=0.25)
    two = sc.upload(hip.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
    one = sc.upload(hip.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
    one.set_option("frames_in_flight", 1)
    n = 6
    outs = [[torch.zeros((sc.height, sc.width, 4), dtype=torch.uint8, device="cuda") for _ in range(n)] for _ in range(2)]
    for k in range(n):
        a, b = copy.deepcopy(sc.desc), copy.deepcopy(sc.desc)
        a.sun = dict(a.sun, rotation=(a.sun["rotation"][0] + 2.0 * k, a.sun["rotation"][1] - 9.0 * k))
        b.sun = dict(b.sun, rotation=(b.sun["rotation"][0] - 1.0 - k, b.sun["rotation"][1] + 5.0 * k + 3.0))
        for r, o in zip((two, one), outs):
            for _ in range(3):
                r.pass_shadow_map(a)                      # main stream, still queued when ...
            r.render_frame_device(b, sc.settings, o[k].data_ptr())   # ... this frame's shadow pass starts on the shadow stream
    two.flush(); one.flush()
    for k in range(n):
        np.testing.assert_array_equal(outs[0][k].cpu().numpy(), outs[1][k].cpu().numpy(), err_msg=f"frame {k}")
    assert any((outs[0][k] != outs[0][k + 1]).any().item() for k in range(n - 1))
    two.close(); one.close()


def test_materials_with_unequal_texture_sizes(pkg, oracle, hip):
    """equal-size triples are stored interleaved; this exercises the other layout (and the fallback 16x16 textures)."""
    rng = np.random.default_rng(21)
    sc = pkg.scenes.config2(scale=0.2)
    mats = []
    for k, (d, n, m) in enumerate(sc.materials):
        if k % 3 == 0:
            mats.append((d, n[::2, ::2].copy(), m[::4, ::2].copy()))          # three different sizes, not square
        elif k % 3 == 1:
            mats.append(pkg.scenes.fallback_textures())                       # white / flat normal / white
        else:
            mats.append((d, n, m))
    sc.materials = mats
    o, r = build_pair(pkg, oracle, hip, sc)
    ref = o.render_frame(sc.desc, sc.settings)
    img = r.render_frame(sc.desc, sc.settings)
    err = np.abs(r.read_output()[0] - o.read_output()[0])
    assert err.max() <= TOL
    assert np.abs(img.astype(np.int16) - ref.astype(np.int16)).max() <= 1
    r.close()


@pytest.mark.parametrize("tm", [0, 1, 2, 5])
def test_post_process_kernel(pkg, oracle, hip, tm):
    """PostProcessPass::run alone (post_process.hlsl:59-93) on an arbitrary HDR image, incl. 0, huge and negative values."""
    rng = np.random.default_rng(tm)
    hdr = np.abs(rng.standard_normal((64, 96, 4)).astype(np.float32)) * rng.choice([0.01, 1.0, 50.0], (64, 96, 1)).astype(np.float32)
    hdr[0, 0] = 0.0
    hdr[0, 1] = (-0.25, 1e6, 1e-8, 1.0)
    r = hip.Renderer(16, 16, 0, 16)
    out, ldr = r.post_process(hdr, (tm, 2.2, 1.3))
    want = np.empty((64, 96, 3), np.float64)
    from test_oracle_kat import tonemap64
    for y in range(64):
        for x in range(96):
            want[y, x] = tonemap64(tm if tm in (1, 2) else 0, hdr[y, x, :3].astype(np.float64), 2.2, 1.3)[1]
    assert np.abs(ldr - want).max() <= TOL
    q = np.clip(np.floor(np.clip(want, 0, 1) * 255 + 0.5), 0, 255)
    assert np.abs(out[..., :3].astype(np.float64) - q).max() <= 1 and (out[..., 3] == 255).all()
    r.close()


def test_full_size_random_gbuffer_properties(pkg, oracle, hip):
    """BASELINE's full 4K size on a RANDOM G-buffer (no rasteriser involved): a stripe against the oracle, culling
    on == off, and a two-shard run == the full run -- properties that do not need the oracle at 8.3 M pixels."""
    sc = pkg.scenes.config3(scale=1.0, tex=128)
    W, H = sc.width, sc.height
    assert (W, H) == (3840, 2160)
    rng = np.random.default_rng(99)
    r = sc.upload(hip.Renderer(W, H, sc.shadow_size, sc.max_lights))
    r.set_option("keep_float_output", 1)
    from importlib import import_module
    lpv = import_module("arctic_renderer_amd.renderer").frame_constants(sc.desc)[1]
    attrs, mat = pkg.scenes.random_gbuffer(rng, H, W, len(sc.materials), coverage=0.97, light_proj_view=lpv)
    smap = (0.3 + 0.5 * rng.random((sc.shadow_size, sc.shadow_size), dtype=np.float32)).astype(np.float32)
    r.write_shadow_map(smap)
    r.write_gbuffer(attrs, mat)
    r.pass_shade(sc.desc, sc.settings)
    ldr, _, rgba = r.read_output(want=("ldr", "rgba8"))
    assert np.isfinite(ldr).all() and (rgba[..., 3] == 255).all()
    assert (rgba[mat == 0xFFFFFFFF][:, :3] == 0).all()                      # no geometry -> black
    # (1) a 12-row stripe against the float64 oracle
    y0 = 1000
    o = sc.upload(oracle.Oracle(W, H, sc.shadow_size, sc.max_lights))
    o.write_shadow_map(smap)
    ref = o.shade_gbuffer(sc.desc, sc.settings, attrs[y0:y0 + 12], mat[y0:y0 + 12], threads=oracle.hardware_threads(), want=("ldr", "rgba8"))
    assert np.abs(ldr[y0:y0 + 12] - ref["ldr"]).max() <= TOL
    assert np.abs(rgba[y0:y0 + 12].astype(np.int16) - ref["rgba8"].astype(np.int16)).max() <= 1
    # (2) the exact culling changes nothing
    r.set_option("culling", 0)
    r.pass_shade(sc.desc, sc.settings)
    assert np.abs(r.read_output(want=("ldr",))[0] - ldr).max() <= 2e-6
    r.set_option("culling", 1)
    r.close()
    # (3) two row shards reproduce the full frame byte for byte
    parts = []
    for b, e in ((0, 1083), (1083, H)):
        s = sc.upload(hip.Renderer(W, H, sc.shadow_size, sc.max_lights, row_begin=b, row_end=e))
        s.write_shadow_map(smap)
        s.write_gbuffer(attrs[b:e], mat[b:e])
        s.pass_shade(sc.desc, sc.settings)
        parts.append(s.read_output(want=("rgba8",))[2])
        s.close()
    np.testing.assert_array_equal(np.concatenate(parts, 0), rgba)


def test_errors_and_state(pkg, hip):
    sc = pkg.scenes.config1(scale=0.25)
    r = hip.Renderer(sc.width, sc.height, 0, 4)
    with pytest.raises(hip.ArcticError) as e:
        r.pass_shade(sc.desc, sc.settings)                       # nothing to shade yet
    assert e.value.code == -4
    with pytest.raises(hip.ArcticError) as e:
        r.create_mesh(sc.meshes[0][0], sc.meshes[0][1], 0)       # material 0 does not exist yet
    assert e.value.code == -1
    sc.upload(r)
    with pytest.raises(hip.ArcticError):
        r.create_mesh(sc.meshes[0][0], sc.meshes[0][1][:4], 0)   # not a triangle list
    lights = pkg.scenes.random_lights(np.random.default_rng(1), 9, (-1, -1, -1), (1, 1, 1))
    r.update_lights(lights)                                      # clamps to max_lights = 4 like the reference clamps to 16
    r.set_option("count_light_evals", 1)
    r.render_frame(sc.desc, sc.settings)
    st = r.stats()
    assert st[6] > 0 and st[5] <= 4 * st[6]                      # at most max_lights evaluations per lit pixel
    r.create_hdri(np.zeros((4, 8, 4), np.float32))               # accepted and ignored (skybox out of scope)
    r.resize(64, 48)
    sc.desc.camera["aspect"] = 64 / 48
    assert r.render_frame(sc.desc, sc.settings).shape == (48, 64, 4)
    r.close()


def test_reference_quantised_mode(pkg, oracle, hip):
    """ps_main's colour rounded through binary16 like the reference's RGBA16F target.  The rounding is a
    discontinuity, so an fp32-level difference may land on the neighbouring half value for a few pixels: the bar is
    <= 1 LSB on RGBA8 with a mismatch-rate bound, and half an fp16 ulp on the float image."""
    sc = pkg.scenes.config3(scale=0.15)
    o, r = build_pair(pkg, oracle, hip, sc)
    o.set_hdr16(1)
    r.set_option("hdr16", 1)
    ref = o.render_frame(sc.desc, sc.settings)
    img = r.render_frame(sc.desc, sc.settings)
    oldr, ohdr, _ = o.read_output()
    hldr, hhdr, _ = r.read_output()
    assert np.array_equal(ohdr, ohdr.astype(np.float16).astype(np.float32))     # the oracle's HDR image is on the fp16 grid
    err = np.abs(hldr - oldr)
    assert np.quantile(err, 0.999) <= TOL and err.max() <= 1e-3
    d = np.abs(img.astype(np.int16) - ref.astype(np.int16))
    assert d.max() <= 1 and (d != 0).mean() < 5e-3
    # and it differs measurably from the fp32 pipeline (the mode is not a no-op)
    r.set_option("hdr16", 0)
    r.pass_shade(sc.desc, sc.settings)
    assert np.abs(r.read_output()[0] - hldr).max() > 1e-5
    r.close()
