"""A/B of the PCF slow path (tiles on a shadow edge): per-lane register window against the per-wave LDS tile (ARCTIC_OPT_DEBUG bit 4),
and the shadow test without the min/max table (bit 3), on config 3 at 4K with 64 and 0 point lights."""
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import __graft_entry__ as e
pkg = e.load_package()
sc = pkg.scenes.CONFIGS[3](scale=1.0)
r = sc.upload(pkg.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
r.pass_shadow_map(sc.desc); r.pass_gbuffer(sc.desc); r.flush()
r.time_shade(sc.desc, sc.settings, warmup=10, iters=10)
for n in (64, 0):
    r.update_lights(sc.lights[:n])
    for rep in range(2):
        for bits, name in ((0, "min/max table + register window (default)"), (16, "min/max table + LDS tile"), (8, "register window only, no table"), (24, "LDS tile only, no table")):
            r.set_option("debug", bits)
            ms = np.median(r.time_shade(sc.desc, sc.settings, warmup=5, iters=40))
            print(f"{n:3d} lights, {name:42s} {ms:.4f} ms", flush=True)
r.set_option("debug", 0)
