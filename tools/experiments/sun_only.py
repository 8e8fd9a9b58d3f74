"""the memory-bound regime of the same pass: config 3's G-buffer with few or no point lights; the two-kernel pass against the
inline paths (ARCTIC_OPT_LIGHT_PATH: the material kernel runs the light loop itself, no stream, no k_light)."""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
import __graft_entry__ as e
pkg = e.load_package()
sc = pkg.scenes.CONFIGS[4](scale=1.0)   # the 4K atrium with 256 lights: prefixes of the list give every count
r = sc.upload(pkg.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
r.pass_shadow_map(sc.desc); r.pass_gbuffer(sc.desc); r.flush()
px = sc.width * sc.height
r.time_shade(sc.desc, sc.settings, warmup=10, iters=10)
for n in [int(a) for a in sys.argv[1:]] or (256, 128, 64, 32, 24, 16, 8, 4, 2, 1, 0):
    r.update_lights(sc.lights[:n])
    row = []
    for inline in (1, 2, 3):
        r.set_option("light_path", inline)
        ms = np.mean(r.time_shade(sc.desc, sc.settings, warmup=5, iters=30))
        row.append(f"{ {1: 'stream', 2: 'inline scalar', 3: 'inline packed'}[inline]} {ms:.4f} ms = {px/ms/1e6:.1f} Gpx/s = {px*80/ms/1e9/8*100:.0f} % of 8 TB/s")
    print(f"{n:3d} point lights: " + "; ".join(row), flush=True)
