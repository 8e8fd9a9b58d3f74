"""ARCTIC_OPT_TEXTURE_TILING: the packed material image in tiles of 4 x 4 texels instead of row-major (needs an MI355X).

The reference creates ONE mip level per texture (src/renderer/rhi.cpp:550) and samples it with MIN_MAG_MIP_LINEAR + WRAP
(src/renderer/forward_pass.cpp:38-51; fetch in shaders/forward.hlsl:98-124): a large texture on a small object is minified at mip 0 and every
pixel's 2 x 2 footprint is its own cache lines.  The tiled layout (common.h TexDesc::tile_row_bytes) changes WHERE a texel lies, nothing else:
the float planes of a frame must be the same bits whatever the layout, for image sizes whose bordered image is not a whole number of tiles, for
footprints that wrap, and in the D3D-style sampler mode.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-4


def render(hip, sc, tiling, sampler=0):
    r = hip.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights)
    r.set_option("texture_tiling", tiling)      # before the materials are created
    sc.upload(r)
    r.set_option("keep_float_output", 1)
    r.set_option("sampler", sampler)
    r.pass_shadow_map(sc.desc); r.pass_gbuffer(sc.desc); r.pass_shade(sc.desc, sc.settings)
    ldr, hdr, rgba = (x.copy() for x in r.read_output())
    frame = r.render_frame(sc.desc, sc.settings).copy()        # the visibility-plane path
    r.close()
    return ldr, hdr, rgba, frame


@pytest.mark.parametrize("cfg,scale,crop", [(2, 0.25, None), (3, 0.12, None), (2, 0.2, (61, 35)), (3, 0.1, (18, 127)), (1, 0.3, (3, 2))],
                         ids=["config2", "config3", "config2-61x35-texels", "config3-18x127-texels", "config1-3x2-texels"])
def test_layout_changes_no_bit(pkg, hip, oracle, cfg, scale, crop):
    sc = pkg.scenes.CONFIGS[cfg](scale=scale)
    if crop:   # image sizes whose bordered image (w + 2) x (h + 2) ends inside a tile, in both directions; all three images of a material stay equal in size (packed)
        sc.materials = [tuple(np.ascontiguousarray(t[:crop[1], :crop[0]]) for t in m) for m in sc.materials]
    rows = [render(hip, sc, tiling) for tiling in (0, 1, -1)]
    for other in rows[1:]:
        for a, b in zip(rows[0], other):
            np.testing.assert_array_equal(a.view(np.uint8), b.view(np.uint8))
    # ... and those bits are the oracle's image
    o = sc.upload(oracle.Oracle(sc.width, sc.height, sc.shadow_size, sc.max_lights))
    o.render_frame(sc.desc, sc.settings)
    assert np.abs(o.read_output()[0] - rows[1][0]).max() <= TOL
    o.close()


def test_tiled_layout_with_the_d3d_style_sampler(pkg, hip):
    sc = pkg.scenes.config2(scale=0.25)
    a, b = render(hip, sc, 0, sampler=5), render(hip, sc, 1, sampler=5)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x.view(np.uint8), y.view(np.uint8))
    plain = render(hip, sc, 1, sampler=0)
    assert np.abs(plain[0] - b[0]).max() > TOL      # (the sampler mode did something)
