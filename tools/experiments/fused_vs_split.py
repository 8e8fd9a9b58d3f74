"""shading pass: two kernels (k_material + k_light) against the single persistent kernel (k_shade_fused)."""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import __graft_entry__ as e
pkg = e.load_package()
for cfg in [int(a) for a in sys.argv[1:]] or (3, 2, 1):
    sc = pkg.scenes.CONFIGS[cfg](scale=1.0)
    r = sc.upload(pkg.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
    r.pass_shadow_map(sc.desc); r.pass_gbuffer(sc.desc); r.flush()
    for culling in (1, 0):
        r.set_option("culling", culling)
        for fused in (0, 1, 0, 1):
            r.set_option("fused", fused)
            ms = r.time_shade(sc.desc, sc.settings, warmup=5, iters=30)
            print(f"config {cfg} culling {culling} fused {fused}: {np.mean(ms):.4f} ms (p10 {np.percentile(ms,10):.4f})", flush=True)
    r.close()
