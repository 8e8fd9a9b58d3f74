import sys, time, numpy as np
sys.path.insert(0,'/root/repo')
import __graft_entry__ as e
pkg=e.load_package()
sc=pkg.scenes.CONFIGS[3](scale=1.0)
def mk(shadow=True):
    r=sc.upload(pkg.Renderer(sc.width,sc.height,sc.shadow_size if shadow else 0,sc.max_lights))
    if shadow: r.pass_shadow_map(sc.desc)
    r.pass_gbuffer(sc.desc); r.flush(); return r
def t(r, label, settings=None):
    ms=r.time_shade(sc.desc, settings or sc.settings, warmup=3, iters=15)
    print(f"{label:40s} {np.median(ms):.4f} ms  (min {ms.min():.4f})", flush=True)
r=mk()
for nb in (1,2,4,8,16):
    r.set_option("bands",nb); t(r,f"full (culling) bands={nb}")
r.set_option("bands",4)
t(r,"full (culling)")
r.set_option("culling",0); t(r,"full (no culling)"); r.set_option("culling",1)
r.update_lights(sc.lights[:0]); t(r,"0 lights")
t(r,"0 lights reinhard",(0,2.2,1.0))
r.write_shadow_map(np.ones((sc.shadow_size,sc.shadow_size),np.float32)); t(r,"0 lights, shadow map all 1.0 (all lit)")
r.update_lights(sc.lights); t(r,"64 lights, all lit"); 
r.update_lights(sc.lights[:16]); t(r,"16 lights, all lit")
r.update_lights(sc.lights[:0])
r.close()
r=mk(False); r.update_lights(sc.lights[:0]); t(r,"0 lights, no shadow map")
# random gbuffer / fallback tiny textures?
r.close()
