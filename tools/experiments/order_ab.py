"""The shading pass's dispatch order, A/B on ONE box inside ONE process (clocks warmed up first): config 3 at 4K, 64 / 16 / 0 point
lights, for a list of (tile_order, tiles_per_wave, order_tail per mille) settings, alternating, medians over the repetitions.
usage: python tools/experiments/order_ab.py [reps] [order,T,tail ...]     e.g.  order_ab.py 3 0,2,60 1,1,60 1,2,60 1,2,0"""
import os, sys, numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import __graft_entry__ as e

args = sys.argv[1:]
reps = int(args[0]) if args and args[0].isdigit() else 3
specs = [tuple(int(x) for x in a.split(",")) for a in args if "," in a] or [(0, 2, 60), (1, 1, 60), (1, 2, 60), (1, 2, 0), (1, 2, 150), (1, 3, 60)]
COUNTS = (64, 16, 0)
pkg = e.load_package()
sc = pkg.scenes.CONFIGS[4](scale=1.0)
r = sc.upload(pkg.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
r.pass_shadow_map(sc.desc); r.pass_gbuffer(sc.desc); r.flush()
r.update_lights(sc.lights[:64])
r.time_shade(sc.desc, sc.settings, warmup=400, iters=10)   # ~80 ms of load: the clocks settle
res = {s: [] for s in specs}
for _ in range(reps):
    for s in specs:
        order, T, tail = s
        r.set_option("tile_order", order); r.set_option("tiles_per_wave", T); r.set_option("order_tail", tail)
        r.pass_gbuffer(sc.desc); r.flush()      # the order belongs to the G-buffer pass
        row = []
        for n in COUNTS:
            r.update_lights(sc.lights[:n])
            row.append(float(np.median(r.time_shade(sc.desc, sc.settings, warmup=20, iters=60))))
        res[s].append(row)
for s in specs:
    a = np.array(res[s])
    print(f"order {s[0]} T {s[1]} tail {s[2]:4d}: " + "  ".join(f"{n} lights {np.median(a[:, i]):.4f} ms ({a[:, i].min():.4f}-{a[:, i].max():.4f})" for i, n in enumerate(COUNTS)), flush=True)
order, classes = r.tile_order() if specs[-1][0] else (None, None)
if classes is not None:
    print(f"costly tiles {int(classes.sum())} of {classes.size} ({classes.mean():.3f})")
r.close()
