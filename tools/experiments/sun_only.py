"""the memory-bound regime of the same pass: config 3's G-buffer with the point lights removed (sun + shadow only)."""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
import __graft_entry__ as e
pkg = e.load_package()
sc = pkg.scenes.CONFIGS[3](scale=1.0)
r = sc.upload(pkg.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
r.pass_shadow_map(sc.desc); r.pass_gbuffer(sc.desc); r.flush()
for n, fused in ((64, 0), (64, 0), (16, 0), (16, 1), (4, 0), (4, 1), (0, 0), (0, 1)):
    r.update_lights(sc.lights[:n])
    r.set_option("fused", fused)
    ms, mm, ml = r.time_shade_split(sc.desc, sc.settings, warmup=5, iters=30)
    px = sc.width * sc.height
    print(f"{n:3d} point lights{' (k_shade_fused)' if fused else ''}: pass {np.mean(ms):.4f} ms (k_material {np.mean(mm):.4f} + k_light {np.mean(ml):.4f}) = {px/np.mean(ms)/1e6:.1f} Gpx/s, "
          f"{px*80/np.mean(ms)/1e9:.2f} TB/s algorithmic = {px*80/np.mean(ms)/1e9/8*100:.0f} % of 8 TB/s", flush=True)
