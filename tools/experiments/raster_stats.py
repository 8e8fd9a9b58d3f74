import sys, numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..' if os.path.basename(os.path.dirname(os.path.abspath(__file__))) == 'tools' else os.path.join('..', '..')))
import __graft_entry__ as e
pkg=e.load_package()
for cfg in (3,2):
    sc=pkg.scenes.CONFIGS[cfg](scale=1.0)
    r=sc.upload(pkg.Renderer(sc.width,sc.height,sc.shadow_size,sc.max_lights))
    r.pass_shadow_map(sc.desc); r.pass_gbuffer(sc.desc); r.flush()
    st=r.stats()
    print(f"config {cfg}: {sc.n_triangles} source triangles; forward: {st[0]} records, {st[1]} 16x16 work items; shadow: {st[2]} records, {st[3]} items", flush=True)
    r.close()
