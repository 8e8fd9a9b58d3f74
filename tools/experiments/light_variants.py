"""k_light variants: lights per loop trip (2 / 4) x workgroups per CU, on the 4K / 64-light pass."""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
import __graft_entry__ as e
pkg = e.load_package()
sc = pkg.scenes.CONFIGS[3](scale=1.0)
r = sc.upload(pkg.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
r.pass_shadow_map(sc.desc); r.pass_gbuffer(sc.desc); r.flush()
for trip in (4, 2, 4, 2):
    for per_cu in (24, 40):
        r.set_option("light_kernel", trip + 16 * per_cu)
        ms, mm, ml = r.time_shade_split(sc.desc, sc.settings, warmup=5, iters=30)
        print(f"lights per trip {trip}, {per_cu} workgroups per CU: k_light {np.mean(ml):.4f} ms (pass {np.mean(ms):.4f})", flush=True)
