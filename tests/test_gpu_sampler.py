"""ARCTIC_OPT_SAMPLER: the D3D-style sampler variants on the HIP side against the oracle in the SAME mode (needs an MI355X).

The reference's texture sampling is its D3D12 sampler's (MIN_MAG_MIP_LINEAR + WRAP, src/renderer/forward_pass.cpp:38-51; the shadow map goes
through the same sampler, shaders/forward.hlsl:84-92; formats src/renderer/renderer.cpp:483-548), and D3D lets the hardware keep texel
coordinates in fixed point with 8 fractional bits.  The oracle has had those variants since round 4 (oracle_set_sampler_mode); round 5 gives the
HIP path the same switch: bit 0 = material footprints, bit 2 = the 25 PCF taps, coordinates snapped to 1/256 texel before the index / weight
split.  Bars as everywhere: float LDR image <= 1e-4 per channel against the float64 oracle -- a flipped PCF tap is 1/25 of a pixel's lit
radiance, far above the gate --, RGBA8 within one step on a small fraction of the channels; the default mode's bytes do not move.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-4
SCENES = [(1, 0.5), (2, 0.25), (3, 0.1), (3, 0.2)]


@pytest.fixture(scope="module", params=SCENES, ids=[f"config{c}-x{s}" for c, s in SCENES])
def pair(request, pkg, oracle, hip):
    cfg, scale = request.param
    sc = pkg.scenes.CONFIGS[cfg](scale=scale)
    o = sc.upload(oracle.Oracle(sc.width, sc.height, sc.shadow_size, sc.max_lights))
    r = sc.upload(hip.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
    r.set_option("keep_float_output", 1)
    o.pass_shadow_map(sc.desc); o.pass_gbuffer(sc.desc)
    r.pass_shadow_map(sc.desc); r.pass_gbuffer(sc.desc)
    r.pass_shade(sc.desc, sc.settings)
    default = [x.copy() for x in r.read_output()]
    yield sc, o, r, default
    r.close()
    o.close()


@pytest.mark.parametrize("mode", [1, 4, 5])
def test_sampler_mode_matches_the_oracle_in_that_mode(pair, mode):
    sc, o, r, default = pair
    if mode == 4 and not sc.shadow_size:
        pytest.skip("config has no shadow map")
    o.set_sampler_mode(mode)
    r.set_option("sampler", mode)
    try:
        o.pass_shade(sc.desc, sc.settings)
        r.pass_shade(sc.desc, sc.settings)
        oldr, _, orgba = o.read_output()
        hldr, _, hrgba = (x.copy() for x in r.read_output())
        err = np.abs(hldr - oldr)
        assert err.max() <= TOL, f"mode {mode}: max err {err.max():.3e} at {np.unravel_index(err.argmax(), err.shape)}"
        d = np.abs(hrgba.astype(np.int16) - orgba.astype(np.int16))
        assert d.max() <= 1 and (d != 0).mean() < 2e-3
        # not vacuous: the mode moves the image by more than the gate somewhere (materials: every config; taps alone: shadow edges)
        moved = np.abs(hldr - default[0]).max()
        assert moved > TOL, f"mode {mode} moved the image by {moved:.2e} only"
        # both light loops, the general tile (debug bit 8 forces it) and the frame path (k_material_vis) give the mode's image too
        for path in (1, 2):
            r.set_option("light_path", path)
            r.pass_shade(sc.desc, sc.settings)
            assert np.abs(r.read_output()[0] - oldr).max() <= TOL
        r.set_option("light_path", 0)
        r.set_option("debug", 256)
        r.pass_shade(sc.desc, sc.settings)
        np.testing.assert_array_equal(r.read_output()[0].view(np.uint32), hldr.view(np.uint32))
        r.set_option("debug", 0)
        frame = r.render_frame(sc.desc, sc.settings)
        np.testing.assert_array_equal(frame, hrgba)
    finally:
        o.set_sampler_mode(0)
        r.set_option("sampler", 0); r.set_option("light_path", 0); r.set_option("debug", 0)
        r.pass_shadow_map(sc.desc); r.pass_gbuffer(sc.desc)   # (render_frame left the visibility plane's frame in place)
    # back in the default mode: the bytes of before
    r.pass_shade(sc.desc, sc.settings)
    for a, b in zip(default, r.read_output()):
        np.testing.assert_array_equal(a, b)


def test_sampler_option_is_validated(pkg, hip):
    sc = pkg.scenes.config1(scale=0.1)
    r = sc.upload(hip.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
    for bad in (2, 3, 8, -1):   # bit 1 (sRGB decode after filtering) exists in the oracle only
        with pytest.raises(hip.ArcticError):
            r.set_option("sampler", bad)
    for ok in (0, 1, 4, 5):
        r.set_option("sampler", ok)
    r.close()


def test_sampler_mode_5_at_4k_on_oracle_stripes(pkg, oracle, hip):
    """the metric's configuration at full size -- 3840 x 2160, the real 4000^2 shadow map (taps 0.4 texel apart: the regime of the 4x4 window
    and the min/max table, whose 8-texel entries must still cover a footprint that snapped across a texel) -- in mode 5, against the oracle in
    mode 5 on stripes through the ceiling, the shadow edges and the sunlit floor"""
    sc = pkg.scenes.config3(scale=1.0)
    r = sc.upload(hip.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
    r.set_option("keep_float_output", 1)
    r.pass_shadow_map(sc.desc); r.pass_gbuffer(sc.desc)
    r.pass_shade(sc.desc, sc.settings)
    default_ldr = r.read_output()[0].copy()
    r.set_option("sampler", 5)
    r.pass_shade(sc.desc, sc.settings)
    h_ldr, _, h_rgba = (x.copy() for x in r.read_output())
    r.set_option("debug", 8)   # the same pass without the shadow min/max table: identical floats (the table stays exact under the snapped taps)
    r.pass_shade(sc.desc, sc.settings)
    np.testing.assert_array_equal(r.read_output()[0].view(np.uint32), h_ldr.view(np.uint32))
    attrs, mat, _, _ = r.read_gbuffer(want=("attrs", "material"))
    o = sc.upload(oracle.Oracle(sc.width, sc.height, sc.shadow_size, sc.max_lights))
    o.write_shadow_map(r.read_shadow_map())
    o.set_sampler_mode(5)
    worst = 0.0
    for y0 in (200, 1100, 1540, 1900):
        ref = o.shade_gbuffer(sc.desc, sc.settings, attrs[y0:y0 + 48], mat[y0:y0 + 48], threads=oracle.hardware_threads(), want=("ldr", "rgba8"))
        err = np.abs(ref["ldr"] - h_ldr[y0:y0 + 48])
        assert err.max() <= TOL, f"rows {y0}..{y0 + 48}: max err {err.max():.3e}"
        d = np.abs(ref["rgba8"].astype(np.int16) - h_rgba[y0:y0 + 48].astype(np.int16))
        assert d.max() <= 1 and (d != 0).mean() < 2e-3
        worst = max(worst, float(err.max()))
    moved = np.abs(h_ldr - default_ldr)
    print(f"config 3 at 4K, sampler mode 5: max |ldr - oracle(mode 5)| = {worst:.2e}; against the default sampler the image moves by up to {moved.max():.3f} "
          f"({float((moved > TOL).mean()) * 100:.2f} % of the channels above 1e-4)")
    assert moved.max() > 10 * TOL
    r.close(); o.close()
