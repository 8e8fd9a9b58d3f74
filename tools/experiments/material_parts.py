"""what k_material's time is made of: timing-only debug bits (bit 0: no texel fetch, bit 1: no shadow test (every pixel lit),
bit 2: no tonemap) on the 4K / 64-light pass.  Images are wrong in these modes."""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
import __graft_entry__ as e
pkg = e.load_package()
sc = pkg.scenes.CONFIGS[3](scale=1.0)
r = sc.upload(pkg.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
r.pass_shadow_map(sc.desc); r.pass_gbuffer(sc.desc); r.flush()
for rep in range(2):
    for dbg in (0, 1, 4, 5):
        r.set_option("debug", dbg)
        ms, mm, ml = r.time_shade_split(sc.desc, sc.settings, warmup=5, iters=30)
        print(f"debug {dbg}: k_material {np.mean(mm):.4f} ms, k_light {np.mean(ml):.4f} ms", flush=True)
