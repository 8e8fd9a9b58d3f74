"""A/B of the prepasses with and without block ownership (ARCTIC_OPT_RASTER_OWNER): per-kernel times come from rocprofv3
(tools/profile_round.sh); this prints whole prepass and whole frame times from HIP events, alternating the two, clocks warm."""
import sys, os, time, copy
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import __graft_entry__ as e
import torch
pkg = e.load_package()
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
sc = pkg.scenes.CONFIGS[cfg](scale=scale)
rs = {}
for owner in (1, 0):
    r = sc.upload(pkg.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
    r.set_option("raster_owner", owner)
    rs[owner] = r
out = torch.empty((sc.height, sc.width, 4), dtype=torch.uint8, device="cuda")
moving = [copy.deepcopy(sc.desc) for _ in range(2)]
moving[1].sun = dict(moving[1].sun, rotation=(moving[1].sun["rotation"][0] - 1.5, moving[1].sun["rotation"][1] + 4.0))
def timed(r, fn, n):   # (the handle's streams are non-blocking: wall clock around flushes, not events on torch's stream)
    for k in range(4): fn(k)
    r.flush(); t = time.perf_counter()
    for k in range(n): fn(k)
    r.flush()
    return (time.perf_counter() - t) / n * 1e3
# warm the clocks
t0 = time.time()
while time.time() - t0 < 0.5:
    rs[1].render_frame_device(sc.desc, sc.settings, out.data_ptr())
rs[1].flush()
for rep in range(3):
    for owner in (1, 0):
        r = rs[owner]
        g = timed(r, lambda k: r.pass_gbuffer(sc.desc), 50)
        s = timed(r, lambda k: r.pass_shadow_map(moving[k & 1]), 50)
        f = timed(r, lambda k: r.render_frame_device(sc.desc, sc.settings, out.data_ptr()), 100)
        m = timed(r, lambda k: r.render_frame_device(moving[k & 1], sc.settings, out.data_ptr()), 100)
        print(f"owner={owner}: pass_gbuffer {g:.4f} ms  pass_shadow_map {s:.4f} ms  frame static {f:.4f} ms  frame moving sun {m:.4f} ms", flush=True)
for r in rs.values():
    r.close()
