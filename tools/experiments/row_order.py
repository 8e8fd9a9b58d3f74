"""ARCTIC_OPT_ROW_ORDER sweep: the shading pass of config 3 (4K atrium) with the row groups of the dispatch visited Q-way
interleaved, Q = 1 .. 32, at 64 / 16 / 4 / 0 point lights.  Back-to-back launches between one pair of events (what bench.py times)
and the median of isolated launches."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import __graft_entry__ as e
pkg = e.load_package()
sc = pkg.scenes.CONFIGS[3](scale=1.0)
r = sc.upload(pkg.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
r.pass_shadow_map(sc.desc); r.pass_gbuffer(sc.desc); r.flush()
px = sc.width * sc.height
r.time_shade(sc.desc, sc.settings, warmup=20, iters=10)
shade = r.prepared_pass_shade(sc.desc, sc.settings)

def back_to_back(n=60):
    for _ in range(10): shade()
    r.flush(); t = time.perf_counter()
    for _ in range(n): shade()
    r.flush()
    return (time.perf_counter() - t) / n * 1e3

orders = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 3, 4, 5]
for rep in range(2):
    for n in (64, 16, 4, 0):
        r.update_lights(sc.lights[:n])
        row = []
        for o in orders:
            r.set_option("row_order", o)
            iso = float(np.median(r.time_shade(sc.desc, sc.settings, warmup=5, iters=30)))
            row.append(f"Q={1 << o}: {back_to_back():.4f} / {iso:.4f}")
        print(f"{n:3d} lights  (back-to-back / isolated ms)  " + "   ".join(row), flush=True)
