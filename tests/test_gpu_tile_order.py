"""The shading pass's dispatch order (round 4, ARCTIC_OPT_TILE_ORDER): arctic_pass_gbuffer leaves a cost class per 8x8 tile next to the
G-buffer -- can a pixel of the tile be lit at all, by the shadow map's min/max table (the shading kernel's own first test,
forward.hlsl:68-96 restated as shade.hip: shadow_quick) -- and k_tile_order turns the classes into the order in which arctic_pass_shade
hands out its strips of 4 tiles.  The order is a scheduling hint: it may change the pass's time, never a byte of its image."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _strip_costly(classes):
    ty, tx = classes.shape
    bpr = (tx + 3) // 4
    pad = np.zeros((ty, bpr * 4), np.uint8)
    pad[:, :tx] = classes
    return pad.reshape(ty, bpr, 4).max(axis=2) != 0        # (ty, bpr)


@pytest.mark.parametrize("cfg,scale", [("config3", 0.13), ("config3", 0.071), ("config2", 0.2)])
def test_dispatch_order_is_placement_only(pkg, hip, cfg, scale):
    """the same bytes with the order off (round 3's geometric order) and on, for every tiles-per-wave and tail setting -- widths whose
    tile count is not a multiple of 4 (a strip with fewer than 4 tiles), pixels without geometry, an environment map"""
    sc = getattr(pkg.scenes, cfg)(scale=scale)
    outs = []
    for order, T, tail in [(0, 0, 60), (1, 0, 60), (1, 1, 0), (1, 2, 1000), (1, 3, 500), (1, 5, 60)]:
        r = sc.upload(hip.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
        if cfg == "config2":
            env = np.ones((8, 16, 4), np.float32); env[:, :, 0] = np.linspace(0.1, 2.0, 16)[None, :]
            r.create_hdri(env)
        r.set_option("tile_order", order); r.set_option("tiles_per_wave", T); r.set_option("order_tail", tail)
        r.pass_shadow_map(sc.desc); r.pass_gbuffer(sc.desc); r.pass_shade(sc.desc, sc.settings)
        outs.append(r.read_output(want=("rgba8",))[2].copy())
        r.close()
    for o in outs[1:]:
        np.testing.assert_array_equal(o, outs[0])


@pytest.mark.parametrize("tail,group", [(0, 0), (60, 0), (400, 0), (60, 2), (250, 3)])
def test_order_is_a_permutation_that_spreads_the_costly_strips_of_every_xcd_list(pkg, hip, tail, group):
    """round 5: eight lists -- list x = the strips of tile rows ty = x (mod 8), the rows the geometric order gives to one XCD -- each ordered on its
    own (costly strips dealt evenly over what the tail leaves, raster order inside both kinds), interleaved in groups of the tiles a wave shades"""
    sc = pkg.scenes.config3(scale=0.13)
    r = sc.upload(hip.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
    r.set_option("tile_order", 1)   # (opt-in since round 5)
    r.set_option("order_tail", tail)
    r.set_option("tiles_per_wave", group)
    r.pass_shadow_map(sc.desc); r.pass_gbuffer(sc.desc)
    order, classes = r.tile_order()
    r.close()
    ty, tx = classes.shape
    bpr = (tx + 3) // 4
    n = bpr * ty
    costly = _strip_costly(classes)
    codes = (np.arange(ty, dtype=np.uint32)[:, None] << 16 | np.arange(bpr, dtype=np.uint32)[None, :])
    assert order.shape == (n,)
    np.testing.assert_array_equal(np.sort(order), np.sort(codes.ravel()))          # every strip exactly once
    assert 0 < int(costly.sum()) < n                                               # the scene has both kinds
    G = group or 1                                                                 # (a frame this small: one tile per wave by default)
    lists = (order >> 16) % 8
    shortest = min(int((lists == x).sum()) for x in range(8))
    whole = 8 * G * (shortest // G)                                                # up to here no list has run out: the slots are all taken
    np.testing.assert_array_equal(lists[:whole], (np.arange(whole) // G) % 8)      # block b = slots b G ..., all of list b % 8
    for x in range(8):
        lst = order[lists == x]                                                    # list x in the order it is dispatched
        n_x = lst.size
        assert n_x == bpr * len(range(x, ty, 8))
        is_costly = costly[lst >> 16, lst & 0xFFFF]
        nl = int(is_costly.sum())
        # the i-th costly strip of the list (raster order) sits at floor(i span / nL), span = what the tail leaves; the cheap ones fill the rest in raster order
        span = max(nl, n_x - n_x * tail // 1000, 1)
        expect = np.zeros(n_x, bool)
        if nl:
            expect[(np.arange(nl, dtype=np.uint64) * span // nl).astype(np.int64)] = True
        np.testing.assert_array_equal(is_costly, expect)
        codes_x, costly_x = codes[x::8].ravel(), costly[x::8].ravel()
        np.testing.assert_array_equal(lst[is_costly], codes_x[costly_x])
        np.testing.assert_array_equal(lst[~is_costly], codes_x[~costly_x])
        if tail and n_x - span > 0:
            assert not is_costly[span:].any()


def test_cost_classes_cover_every_pixel_that_can_be_lit(pkg, hip, oracle):
    """conservative: a tile holding a covered pixel whose PCF result (oracle: calculate_shadow, forward.hlsl:68-96) is below 1 is costly --
    and the hint is not vacuous: tiles in full shadow exist and are cheap"""
    sc = pkg.scenes.config3(scale=0.1)
    r = sc.upload(hip.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
    r.set_option("tile_order", 1)
    r.pass_shadow_map(sc.desc); r.pass_gbuffer(sc.desc)
    _, classes = r.tile_order()
    attrs, mat, _, _ = r.read_gbuffer()
    smap = r.read_shadow_map()
    r.close()
    h, w = mat.shape
    lit_tile = np.zeros_like(classes, bool)
    full_shadow_tile = np.ones_like(classes, bool)
    for y in range(h):
        for x in range(w):
            if mat[y, x] == 0xFFFFFFFF:
                continue
            s = oracle.calculate_shadow(smap, attrs[y, x, 14:18])
            if s < 1.0:
                lit_tile[y // 8, x // 8] = True
                full_shadow_tile[y // 8, x // 8] = False
    assert (classes[lit_tile] == 1).all()
    assert lit_tile.any() and (classes == 0).any()
    # tight enough to be worth having: most tiles in full shadow are recognised (the table is conservative at shadow edges only)
    assert (classes[full_shadow_tile] == 0).mean() > 0.5


def test_order_of_an_interleaved_shard(pkg, hip):
    """a shard builds the order of its own tile rows: its rows equal the whole frame's"""
    sc = pkg.scenes.config3(scale=0.13)
    full = sc.upload(hip.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
    full.set_option("tile_order", 0)
    full.pass_shadow_map(sc.desc); full.pass_gbuffer(sc.desc); full.pass_shade(sc.desc, sc.settings)
    ref = full.read_output(want=("rgba8",))[2].copy()
    full.close()
    world, band = 3, 16
    for rank in range(world):
        r = sc.upload(hip.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights, band_rows=band, shard=(rank, world)))
        r.set_option("tile_order", 1)
        r.pass_shadow_map(sc.desc); r.pass_gbuffer(sc.desc); r.pass_shade(sc.desc, sc.settings)
        order, classes = r.tile_order()
        assert order.size == ((classes.shape[1] + 3) // 4) * classes.shape[0]
        got = r.read_output(want=("rgba8",))[2]
        rows = [y for y in range(sc.height) if (y // band) % world == rank]
        np.testing.assert_array_equal(got, ref[rows])
        r.close()
