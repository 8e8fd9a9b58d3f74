"""BASELINE.json's configurations at their FULL sizes, HIP against the CPU oracle through the C-ABI (needs an MI355X and its
host's cores: the oracle rasterises a 4K frame in seconds and shades it on all hardware threads).

The scaled scenes of test_gpu_parity.py shrink the shadow map with the frame (S = 400 at x0.1), so the PCF taps there are
0.04 texel apart; here the maps have their real size (2048 / 4000 / 4096: taps 0.2 / 0.4 / 0.41 texel apart -- the regime the
4x4 window and the min/max table of shade.hip are built for, forward.hlsl:68-96 with shadow_map_pass.hpp:23's 4000).

Bars as everywhere: shadow map, visibility and all 18 G-buffer attributes bit for bit; float LDR image <= 1e-4 per channel.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-4


def prepass_pair(pkg, oracle, hip, sc):
    o = sc.upload(oracle.Oracle(sc.width, sc.height, sc.shadow_size, sc.max_lights))
    r = sc.upload(hip.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
    r.set_option("keep_float_output", 1)
    o.pass_shadow_map(sc.desc); o.pass_gbuffer(sc.desc)
    r.pass_shadow_map(sc.desc); r.pass_gbuffer(sc.desc)
    return o, r


def assert_prepass_bit_exact(sc, o, r):
    if sc.shadow_size:
        a, b = o.read_shadow_map(), r.read_shadow_map()
        assert (a < 1.0).mean() > 0.001   # config 2: the models cover half a percent of the +-16 m light window
        np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))
    oa, om, od, ot = o.read_gbuffer()
    ha, hm, hd, ht = r.read_gbuffer()
    np.testing.assert_array_equal(om, hm)
    np.testing.assert_array_equal(ot, ht)
    np.testing.assert_array_equal(od.view(np.uint32), hd.view(np.uint32))
    np.testing.assert_array_equal(oa.view(np.uint32), ha.view(np.uint32))     # all 18 attributes
    return oa, om


def assert_image_parity(o_ldr, o_rgba, h_ldr, h_rgba):
    err = np.abs(o_ldr - h_ldr)
    assert err.max() <= TOL, f"max |ldr - oracle| = {err.max():.3e}"
    d = np.abs(o_rgba.astype(np.int16) - h_rgba.astype(np.int16))
    assert d.max() <= 1 and (d != 0).mean() < 2e-3
    return float(err.max())


def test_config3_4k_frame_whole_frame_against_oracle(pkg, oracle, hip):
    """the headline configuration end to end: 3840x2160 through the rasteriser, the real 4000^2 shadow map, 64 point lights,
    ACES -- every pixel of the frame against the float64 oracle; then the frame path (visibility plane) gives the same bytes."""
    sc = pkg.scenes.config3(scale=1.0)
    assert (sc.width, sc.height, sc.shadow_size, len(sc.lights)) == (3840, 2160, 4000, 64)
    o, r = prepass_pair(pkg, oracle, hip, sc)
    assert_prepass_bit_exact(sc, o, r)
    o.pass_shade(sc.desc, sc.settings, threads=oracle.hardware_threads())
    r.pass_shade(sc.desc, sc.settings)
    o_ldr, _, o_rgba = o.read_output()
    h_ldr, _, h_rgba = (x.copy() for x in r.read_output())
    worst = assert_image_parity(o_ldr, o_rgba, h_ldr, h_rgba)
    print(f"config 3 at 4K: max |ldr - oracle| = {worst:.2e}, lit fraction {float((o_ldr.sum(-1) > 0).mean()):.3f}")
    # partially shadowed pixels (PCF results strictly between 0 and 1) exist: the 25-tap path was exercised at S = 4000
    for path in (1, 2):
        r.set_option("light_path", path)
        r.pass_shade(sc.desc, sc.settings)
        assert np.abs(r.read_output()[0] - o_ldr).max() <= TOL
    r.set_option("light_path", 0)
    frame = r.render_frame(sc.desc, sc.settings)
    np.testing.assert_array_equal(frame, h_rgba)
    r.set_option("debug", 8)   # the same pass without the shadow min/max table: identical floats
    r.pass_shade(sc.desc, sc.settings)
    np.testing.assert_array_equal(r.read_output()[0].view(np.uint32), h_ldr.view(np.uint32))
    r.close(); o.close()


def test_config1_512_no_shadow_map_against_oracle(pkg, oracle, hip):
    """config 1 at its real 512 x 512 (the reference's own CPU-runnable case: one sphere, one directional light, no shadow map, Reinhard):
    prepass bit for bit, every pixel of the image against the float64 oracle, and the frame path gives the same bytes"""
    sc = pkg.scenes.config1(scale=1.0)
    assert (sc.width, sc.height, sc.shadow_size, len(sc.lights)) == (512, 512, 0, 0)
    o, r = prepass_pair(pkg, oracle, hip, sc)
    assert_prepass_bit_exact(sc, o, r)
    o.pass_shade(sc.desc, sc.settings, threads=oracle.hardware_threads())
    r.pass_shade(sc.desc, sc.settings)
    o_ldr, _, o_rgba = o.read_output()
    h_ldr, _, h_rgba = (x.copy() for x in r.read_output())
    assert_image_parity(o_ldr, o_rgba, h_ldr, h_rgba)
    np.testing.assert_array_equal(r.render_frame(sc.desc, sc.settings), h_rgba)
    r.close(); o.close()


def test_config2_1080p_2048_shadow_against_oracle(pkg, oracle, hip):
    """config 2 at its real size: 1920x1080, 2048^2 shadow map, sun only, Reinhard; 40 % of the frame is background."""
    sc = pkg.scenes.config2(scale=1.0)
    assert (sc.width, sc.height, sc.shadow_size) == (1920, 1080, 2048)
    o, r = prepass_pair(pkg, oracle, hip, sc)
    assert_prepass_bit_exact(sc, o, r)
    o.pass_shade(sc.desc, sc.settings, threads=oracle.hardware_threads())
    r.pass_shade(sc.desc, sc.settings)
    o_ldr, _, o_rgba = o.read_output()
    h_ldr, _, h_rgba = r.read_output()
    assert_image_parity(o_ldr, o_rgba, h_ldr, h_rgba)
    r.close(); o.close()


def test_config4_4k_256_lights(pkg, oracle, hip):
    """config 4 at its real size: 3840x2160, 256 point lights (forward.hlsl:224-231 far beyond renderer.hpp:22's cap of 16), the
    4000^2 map.  Prepass bit for bit over the whole frame; three 64-row stripes against the float64 oracle (the whole frame at 257
    light evaluations per pixel is minutes of CPU time); exact culling on == off; and the sharding config 4 is defined with: the
    frame rendered as 2 / 4 / 8 interleaved 16-row-band shards (one handle each, all on this device) and put together by
    arctic_assemble_frame -- the root's half of arctic_gather_frame -- must be the single-device frame byte for byte."""
    import torch
    sc = pkg.scenes.config4(scale=1.0)
    assert (sc.width, sc.height, sc.shadow_size, len(sc.lights)) == (3840, 2160, 4000, 256)
    o, r = prepass_pair(pkg, oracle, hip, sc)
    attrs, mat = assert_prepass_bit_exact(sc, o, r)
    r.pass_shade(sc.desc, sc.settings)
    h_ldr, _, h_rgba = (x.copy() for x in r.read_output())
    assert np.isfinite(h_ldr).all() and (h_rgba[..., 3] == 255).all()
    for y0 in (300, 1050, 1850):   # ceiling / far wall, the middle of the atrium, the sunlit floor
        ref = o.shade_gbuffer(sc.desc, sc.settings, attrs[y0:y0 + 64], mat[y0:y0 + 64], threads=oracle.hardware_threads(), want=("ldr", "rgba8"))
        worst = assert_image_parity(ref["ldr"], ref["rgba8"], h_ldr[y0:y0 + 64], h_rgba[y0:y0 + 64])
        print(f"config 4 at 4K, rows {y0}..{y0 + 64}: max |ldr - oracle| = {worst:.2e}")
    del attrs, mat
    o.close()
    r.set_option("culling", 0)     # every pixel through the 256-light loop: same image
    r.pass_shade(sc.desc, sc.settings)
    assert np.abs(r.read_output()[0] - h_ldr).max() <= 2e-6
    r.set_option("culling", 1)
    frame = r.render_frame(sc.desc, sc.settings)   # the frame path from the visibility plane: same bytes
    np.testing.assert_array_equal(frame, h_rgba)
    for world in (2, 4, 8):
        parts = []
        for k in range(world):
            s = sc.upload(hip.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights, band_rows=16, shard=(k, world)))
            parts.append(torch.as_tensor(s.render_frame(sc.desc, sc.settings).reshape(-1), device="cuda"))
            s.close()
        assert sum(p.numel() for p in parts) == sc.width * sc.height * 4     # 135 bands: unequal shards at 2, 4 and 8 ranks
        staging = torch.cat(parts)
        out = torch.zeros((sc.height, sc.width, 4), dtype=torch.uint8, device="cuda")
        a = sc.upload(hip.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights, band_rows=16, shard=(world - 1, world)))
        a.assemble_frame(staging.data_ptr(), out.data_ptr(), world)
        a.flush()
        a.close()
        np.testing.assert_array_equal(out.cpu().numpy(), h_rgba)
    r.close()


def test_config5_8k_prepass_bitwise_and_oracle_stripes(pkg, oracle, hip):
    """config 5 at 7680x4320 with 1024 point lights and the 4096^2 map: shadow map and G-buffer bit for bit over all 33 M
    pixels; the shaded image against the oracle on stripes (64 rows each) through lit and shadowed regions -- the whole 8K
    frame with 1025 light evaluations per pixel is minutes of CPU time -- and the whole frame by properties."""
    sc = pkg.scenes.config5(scale=1.0)
    assert (sc.width, sc.height, sc.shadow_size, len(sc.lights)) == (7680, 4320, 4096, 1024)
    o, r = prepass_pair(pkg, oracle, hip, sc)
    attrs, mat = assert_prepass_bit_exact(sc, o, r)
    r.pass_shade(sc.desc, sc.settings)
    h_ldr, _, h_rgba = (x.copy() for x in r.read_output())
    assert np.isfinite(h_ldr).all() and (h_rgba[..., 3] == 255).all()
    for y0 in (600, 2100, 3700):   # ceiling / far wall, the middle of the atrium, the sunlit floor
        ref = o.shade_gbuffer(sc.desc, sc.settings, attrs[y0:y0 + 64], mat[y0:y0 + 64], threads=oracle.hardware_threads(), want=("ldr", "rgba8"))
        worst = assert_image_parity(ref["ldr"], ref["rgba8"], h_ldr[y0:y0 + 64], h_rgba[y0:y0 + 64])
        print(f"config 5 at 8K, rows {y0}..{y0 + 64}: max |ldr - oracle| = {worst:.2e}")
    del attrs
    r.set_option("culling", 0)     # every pixel through the light loop: same image
    r.pass_shade(sc.desc, sc.settings)
    assert np.abs(r.read_output()[0] - h_ldr).max() <= 2e-6
    r.set_option("culling", 1)
    frame = r.render_frame(sc.desc, sc.settings)   # and the frame path from the visibility plane: same bytes
    np.testing.assert_array_equal(frame, h_rgba)
    r.close(); o.close()
