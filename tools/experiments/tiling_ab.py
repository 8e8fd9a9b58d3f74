"""Row-major against 4 x 4-tiled material images (ARCTIC_OPT_TEXTURE_TILING) on one box, two handles of the same scene alternating with the clocks
warmed up: the shading pass of a BASELINE configuration at a few light counts.   usage: python tools/experiments/tiling_ab.py [config=3]"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import __graft_entry__ as e
pkg = e.load_package()
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
sc = pkg.scenes.CONFIGS[cfg](scale=1.0)
hs = {}
for t in (0, 1):
    r = pkg.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights)
    r.set_option("texture_tiling", t)
    sc.upload(r)
    r.pass_shadow_map(sc.desc); r.pass_gbuffer(sc.desc); r.flush()
    hs[t] = r
hs[0].time_shade(sc.desc, sc.settings, warmup=300, iters=10)
out = {}
counts = [n for n in (64, 16, 0) if n <= len(sc.lights)] or [0]
for n in counts:
    res = {0: [], 1: []}
    for rep in range(4):
        for t in (0, 1):
            hs[t].update_lights(sc.lights[:n])
            res[t].append(float(np.median(hs[t].time_shade(sc.desc, sc.settings, warmup=10, iters=40))))
    out[f"{n}_lights_ms"] = {"row_major": round(float(np.median(res[0])), 4), "tiled_4x4": round(float(np.median(res[1])), 4)}
print(json.dumps({"config": cfg, **out}))
