"""What one rank of R pays per whole frame (its prepass is not 1/R of the frame's): arctic_render_frame_device on ONE shard handle --
interleaved bands of 16 rows (what bench.py --gpus N uses) and contiguous row ranges -- static sun and with the shadow map redrawn.
No exchange, one GPU: the per-rank fixed cost of DESIGN.md section 5.   usage: python tools/experiments/shard_cost.py [R ...]"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import __graft_entry__ as e
pkg = e.load_package()
sc = pkg.scenes.CONFIGS[3](scale=1.0)
def frames(r):
    out = []
    for cache in (1, 0):
        r.set_option("shadow_cache", cache)
        for i in range(60): r.render_frame_device(sc.desc, sc.settings, None)
        r.flush(); t = time.perf_counter()
        for i in range(40): r.render_frame_device(sc.desc, sc.settings, None)
        r.flush(); out.append((time.perf_counter() - t) / 40 * 1e3)
    return out
full = sc.upload(pkg.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
for i in range(200): full.render_frame_device(sc.desc, sc.settings, None)
f = frames(full); full.close()
print(f"whole frame on one device: {f[0]:.4f} ms static sun, {f[1]:.4f} ms with shadow redraw", flush=True)
for R in [int(a) for a in sys.argv[1:]] or (2, 4, 8):
    for kind in ("bands", "rows"):
        if kind == "bands":
            r = sc.upload(pkg.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights, band_rows=16, shard=(R // 2, R)))
        else:
            h = (sc.height // R + 7) // 8 * 8
            r = sc.upload(pkg.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights, row_begin=h * (R // 2), row_end=min(sc.height, h * (R // 2 + 1))))
        t = frames(r); st = r.stats(); r.close()
        print(f"R = {R}, rank {R // 2}, {kind:5s}: {t[0]:.4f} ms static sun ({t[0] / f[0]:.2f} of the whole frame; ideal {1 / R:.2f}), {t[1]:.4f} ms with shadow redraw; "
              f"records / items of its last forward prepass: {int(st[0])} / {int(st[1])}", flush=True)
