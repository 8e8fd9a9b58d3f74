"""light-count sweep of the shading pass on config 3's G-buffer (the 4K atrium): the scalar loop against the packed loop
(ARCTIC_OPT_LIGHT_PATH 1 / 2), Gpx/s and the fraction of the 8 TB/s roof on the algorithmic 80 B/pixel."""
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import __graft_entry__ as e
pkg = e.load_package()
sc = pkg.scenes.CONFIGS[4](scale=1.0)   # the 4K atrium with 256 lights: prefixes of the list give every count
r = sc.upload(pkg.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
r.pass_shadow_map(sc.desc); r.pass_gbuffer(sc.desc); r.flush()
px = sc.width * sc.height
r.time_shade(sc.desc, sc.settings, warmup=10, iters=10)
for n in [int(a) for a in sys.argv[1:]] or (256, 64, 32, 16, 8, 4, 1, 0):
    r.update_lights(sc.lights[:n])
    row = []
    for path in (1, 2):
        r.set_option("light_path", path)
        ms = np.median(r.time_shade(sc.desc, sc.settings, warmup=5, iters=40))
        row.append(f"{ {1: 'scalar', 2: 'packed'}[path]} {ms:.4f} ms = {px/ms/1e6:.1f} Gpx/s = {px*80/ms/1e9/8*100:.0f} % of 8 TB/s")
    print(f"{n:3d} point lights: " + "; ".join(row), flush=True)
