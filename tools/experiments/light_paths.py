"""scalar light loop (ARCTIC_OPT_LIGHT_PATH 1) against the packed one (2) over the light count, warm clocks: where the automatic choice should switch."""
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import __graft_entry__ as e
pkg = e.load_package()
sc = pkg.scenes.CONFIGS[4](scale=1.0)
r = sc.upload(pkg.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
r.pass_shadow_map(sc.desc); r.pass_gbuffer(sc.desc); r.flush()
r.update_lights(sc.lights[:64])
r.time_shade(sc.desc, sc.settings, warmup=300, iters=10)
for rep in range(2):
    for n in (32, 16, 12, 8, 4, 2, 1, 0):
        r.update_lights(sc.lights[:n])
        row = []
        for path in (1, 2):
            r.set_option("light_path", path)
            row.append(f"{ {1: 'scalar', 2: 'packed'}[path]} {np.median(r.time_shade(sc.desc, sc.settings, warmup=20, iters=60)):.4f}")
        print(f"{n:3d} lights (ms): " + "  ".join(row), flush=True)
