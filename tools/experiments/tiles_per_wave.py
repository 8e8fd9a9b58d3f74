"""ARCTIC_OPT_TILES_PER_WAVE x ARCTIC_OPT_ROW_ORDER sweep of the shading pass on config 3's G-buffer (4K), a few light counts."""
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import __graft_entry__ as e
pkg = e.load_package()
sc = pkg.scenes.CONFIGS[4](scale=1.0)
r = sc.upload(pkg.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
r.pass_shadow_map(sc.desc); r.pass_gbuffer(sc.desc); r.flush()
r.time_shade(sc.desc, sc.settings, warmup=10, iters=10)
Ts = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4, 6, 8, 12, 18]
for n in (64, 16, 0):
    r.update_lights(sc.lights[:n])
    r.set_option("light_path", 2)
    for order in (0, 1):
        r.set_option("row_order", order)
        row = []
        for T in Ts:
            r.set_option("tiles_per_wave", T)
            row.append(f"T={T}: {np.median(r.time_shade(sc.desc, sc.settings, warmup=5, iters=40)):.4f}")
        print(f"{n:3d} lights, row order {order} (ms)  " + "  ".join(row), flush=True)
