"""do k_material (memory bound) and k_light (VALU bound) of two independent passes overlap when they sit on two HIP
streams?  Two handles, each on its own stream, same scene; alternate pass_shade calls and compare with one handle."""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import __graft_entry__ as e
pkg = e.load_package()
sc = pkg.scenes.CONFIGS[3](scale=1.0)
hs = []
for i in range(2):
    r = sc.upload(pkg.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
    r.pass_shadow_map(sc.desc); r.pass_gbuffer(sc.desc); r.flush()
    hs.append(r)
def run(handles, n):
    shade = [h.prepared_pass_shade(sc.desc, sc.settings) for h in handles]
    for s in shade: s(None)
    for h in handles: h.flush()
    t = time.perf_counter()
    for i in range(n):
        for s in shade: s(None)
    for h in handles: h.flush()
    return (time.perf_counter() - t) / (n * len(handles)) * 1e3
for rep in range(3):
    print(f"one handle : {run(hs[:1], 200):.4f} ms per pass", flush=True)
    print(f"two handles: {run(hs, 100):.4f} ms per pass (own streams, alternating launches)", flush=True)
