import sys, numpy as np
sys.path.insert(0,'/root/repo')
import __graft_entry__ as e
pkg=e.load_package()
sc=pkg.scenes.CONFIGS[3](scale=1.0)
r=sc.upload(pkg.Renderer(sc.width,sc.height,sc.shadow_size,sc.max_lights))
r.pass_shadow_map(sc.desc); r.pass_gbuffer(sc.desc); r.flush()
r.set_option("count_light_evals",1)
r.pass_shade(sc.desc, sc.settings); r.flush()
st=r.stats(); npx=sc.width*sc.height
print("stats", st, "lit px", st[6], "frac", st[6]/npx, "evals", st[5], "evals per lit px", st[5]/max(st[6],1), "evals per px", st[5]/npx)
