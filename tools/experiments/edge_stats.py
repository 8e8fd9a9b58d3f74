"""Shadow-edge statistics of the fast tile (arctic_stats [12..15], the counting variant of the shading kernel) for a BASELINE configuration at full
size: how many tiles the shadow map's min/max table leaves undecided, how many of their pixels, and how many run the 25 PCF compares.
usage: python tools/experiments/edge_stats.py [config=3]"""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import __graft_entry__ as e
pkg = e.load_package()
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
sc = pkg.scenes.CONFIGS[cfg](scale=1.0)
r = sc.upload(pkg.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
r.pass_shadow_map(sc.desc); r.pass_gbuffer(sc.desc); r.flush()
r.set_option("count_light_evals", 1)
r.pass_shade(sc.desc, sc.settings)
st = [int(x) for x in r.stats()]
tiles = ((sc.width + 7) // 8) * ((sc.height + 7) // 8)
print(json.dumps({"config": cfg, "tiles": tiles, "lit_tiles": st[9], "lit_pixels": st[6], "edge_tiles": st[12], "undecided_pixels": st[13],
                  "tiles_with_25_taps": st[14], "pixels_with_25_taps": st[15],
                  "undecided_pixels_per_edge_tile": round(st[13] / max(st[12], 1), 2), "tap_pixels_per_tap_tile": round(st[15] / max(st[14], 1), 2)}))
r.close()
