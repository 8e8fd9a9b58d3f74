import sys, os, numpy as np
sys.path.insert(0,'/root/repo')
import __graft_entry__ as e
pkg=e.load_package()
sc=pkg.scenes.CONFIGS[3](scale=1.0)
r=sc.upload(pkg.Renderer(sc.width,sc.height,sc.shadow_size,sc.max_lights))
r.pass_shadow_map(sc.desc); r.pass_gbuffer(sc.desc); r.flush()
for nb in (1,4,8):
    r.set_option("bands",nb)
    ms=r.time_shade(sc.desc, sc.settings, warmup=3, iters=15)
    print(f"A_LDS_KB={os.environ.get('ARCTIC_A_LDS_KB')} bands={nb}: {np.median(ms):.4f} ms", flush=True)
