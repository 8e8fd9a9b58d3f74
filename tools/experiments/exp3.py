import sys, os, numpy as np
sys.path.insert(0,'/root/repo')
import __graft_entry__ as e
pkg=e.load_package()
sc=pkg.scenes.CONFIGS[3](scale=1.0)
r=sc.upload(pkg.Renderer(sc.width,sc.height,sc.shadow_size,sc.max_lights))
r.pass_shadow_map(sc.desc); r.pass_gbuffer(sc.desc); r.flush()
for rep in range(2):
  for trip in (4,2):
    for bpc in (8,12,16,24,32,48):
        r.set_option("light_kernel", trip + 16*bpc)
        ms,a,b=r.time_shade_split(sc.desc, sc.settings, warmup=3, iters=15)
        print(f"trip={trip} blocks/CU={bpc}: pass {np.median(ms):.4f}  k_material {np.median(a):.4f}  k_light {np.median(b):.4f}", flush=True)
