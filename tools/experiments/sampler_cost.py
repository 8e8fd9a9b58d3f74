"""What ARCTIC_OPT_SAMPLER costs: the shading pass of config 3 at 4K with the default sampler and with the D3D-style variants (bit 0 material
footprints, bit 2 PCF taps), alternating in ONE process with the clocks warmed up.   usage: python tools/experiments/sampler_cost.py"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import __graft_entry__ as e
pkg = e.load_package()
sc = pkg.scenes.CONFIGS[3](scale=1.0)
r = sc.upload(pkg.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
r.pass_shadow_map(sc.desc); r.pass_gbuffer(sc.desc); r.flush()
r.time_shade(sc.desc, sc.settings, warmup=300, iters=10)
out = {}
for n in (64, 16, 0):
    r.update_lights(sc.lights[:n])
    res = {m: [] for m in (0, 1, 4, 5)}
    for rep in range(3):
        for m in res:
            r.set_option("sampler", m)
            res[m].append(float(np.median(r.time_shade(sc.desc, sc.settings, warmup=10, iters=40))))
    out[f"{n}_lights_ms"] = {f"mode_{m}": round(float(np.median(v)), 4) for m, v in res.items()}
print(json.dumps(out))
r.close()
