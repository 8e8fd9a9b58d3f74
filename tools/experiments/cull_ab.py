"""ARCTIC_OPT_CLUSTER_CULL off / k_setup only / k_setup + k_vertex: whole frames (static sun, shadow map redrawn) on one device and on one
rank of R (interleaved 16-row bands, contiguous rows), handles alternating in one process.   usage: python tools/experiments/cull_ab.py [reps]"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import __graft_entry__ as e
pkg = e.load_package()
sc = pkg.scenes.CONFIGS[int(os.environ.get("CONFIG", "3"))](scale=1.0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
def frames(r, cache, n=60):
    r.set_option("shadow_cache", cache)
    for i in range(20): r.render_frame_device(sc.desc, sc.settings, None)
    r.flush(); t = time.perf_counter()
    for i in range(n): r.render_frame_device(sc.desc, sc.settings, None)
    r.flush(); return (time.perf_counter() - t) / n * 1e3
def world(label, **kw):
    hs = []
    for cull in (0, 1, 3):
        r = sc.upload(pkg.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights, **kw))
        r.set_option("cluster_cull", cull)
        for i in range(100): r.render_frame_device(sc.desc, sc.settings, None)
        hs.append(r)
    res = np.zeros((3, 2, reps))
    for k in range(reps):
        for j, r in enumerate(hs):
            res[j, 0, k] = frames(r, 1); res[j, 1, k] = frames(r, 0)
    hs[2].set_option("debug", 1024); hs[2].set_option("shadow_cache", 0); hs[2].render_frame_device(sc.desc, sc.settings, None); hs[2].flush()
    c = [hs[2].cull_counts(False), hs[2].cull_counts(True) if sc.shadow_size else None]
    for r in hs: r.close()
    m = np.median(res, axis=2)
    print(f"{label}: static sun {m[0, 0]:.4f} -> {m[1, 0]:.4f} -> {m[2, 0]:.4f} ms   shadow redrawn {m[0, 1]:.4f} -> {m[1, 1]:.4f} -> {m[2, 1]:.4f} ms   (off -> k_setup -> k_setup + k_vertex; "
          f"forward pass skips {c[0][1]} of {c[0][0]} clusters, {c[0][3]} of {c[0][2]} vertex blocks" + (f"; shadow pass {c[1][1]} of {c[1][0]}, {c[1][3]} of {c[1][2]})" if c[1] is not None else ")"), flush=True)
world("whole frame")
for R in (2, 4, 8):
    world(f"one rank of {R}, bands of 16 rows", band_rows=16, shard=(R // 2, R))
    h = (sc.height // R + 7) // 8 * 8
    world(f"one rank of {R}, rows {h * (R // 2)}..{min(sc.height, h * (R // 2 + 1))}", row_begin=h * (R // 2), row_end=min(sc.height, h * (R // 2 + 1)))
