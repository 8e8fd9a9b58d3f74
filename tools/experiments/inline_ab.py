"""A/B/C on the headline workload (4K, config 3, 64 lights): the three light paths (ARCTIC_OPT_LIGHT_PATH), interleaved repetitions."""
import sys, os, numpy as np
sys.path.insert(0, '/root/repo')
import __graft_entry__ as e
pkg = e.load_package()
sc = pkg.scenes.CONFIGS[3](scale=1.0)
r = sc.upload(pkg.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
r.pass_shadow_map(sc.desc); r.pass_gbuffer(sc.desc); r.flush()
r.time_shade(sc.desc, sc.settings, warmup=20, iters=10)
res = {1: [], 2: [], 3: []}
for rep in range(6):
    for mode in (1, 2, 3):
        r.set_option("light_path", mode)
        res[mode].append(np.mean(r.time_shade(sc.desc, sc.settings, warmup=5, iters=40)))
for mode, name in ((1, "k_material -> stream -> k_light"), (2, "inline, scalar loop"), (3, "inline, packed pairs")):
    print(f"{name:32s}:", " ".join(f"{x:.4f}" for x in res[mode]), "ms; mean of the last five", f"{np.mean(res[mode][1:]):.4f}")
