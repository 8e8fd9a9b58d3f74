"""ARCTIC_OPT_SMALL_TRIANGLES off / on: the shadow pass alone (back to back) and whole frames with the shadow map redrawn every frame, handles alternating in
one process; prints a checksum of the map per handle.  The library is chosen with ARCTIC_HIP_LIBRARY (builds with another SMALL_PX: build_variant.sh, GEO=1).
usage: python tools/experiments/small_ab.py [reps]   (CONFIG=3 by default; TRACE=1: only 30 + 30 shadow passes, for rocprofv3 --kernel-trace --stats)"""
import hashlib, os, sys, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import __graft_entry__ as e
pkg = e.load_package()
sc = pkg.scenes.CONFIGS[int(os.environ.get("CONFIG", "3"))](scale=1.0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
lib = os.path.basename(os.environ.get("ARCTIC_HIP_LIBRARY", "default"))
def handle(small, owner=-1):
    r = sc.upload(pkg.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
    r.set_option("small_triangles", small); r.set_option("shadow_cache", 0); r.set_option("raster_owner", owner)
    for i in range(10): r.pass_shadow_map(sc.desc)
    r.flush()
    return r
def shadow_ms(r, n=40):
    r.flush(); t = time.perf_counter()
    for i in range(n): r.pass_shadow_map(sc.desc)
    r.flush(); return (time.perf_counter() - t) / n * 1e3
def frame_ms(r, cache, n=60):
    r.set_option("shadow_cache", cache)
    for i in range(20): r.render_frame_device(sc.desc, sc.settings, None)
    r.flush(); t = time.perf_counter()
    for i in range(n): r.render_frame_device(sc.desc, sc.settings, None)
    r.flush(); return (time.perf_counter() - t) / n * 1e3
# MODES="small:owner,..." (ARCTIC_OPT_SMALL_TRIANGLES : ARCTIC_OPT_RASTER_OWNER; default off and on with the library's choice of owners); the summary line names them in order
modes = [tuple(int(x) for x in m.split(":")) for m in os.environ.get("MODES", "0:-1,1:-1").split(",")]
hs = [handle(*m) for m in modes]
if os.environ.get("TRACE"):
    for r in hs:
        for i in range(30): r.pass_shadow_map(sc.desc)
        r.flush()
    sys.exit(0)
for r, name in zip(hs, (f"{m[0]} (owners {m[1]})" for m in modes)):
    st = [int(x) for x in r.stats()[:4]]
    print(f"{lib} small triangles {name}: shadow records {st[2]}, work items {st[3]}, map sha1 {hashlib.sha1(r.read_shadow_map().tobytes()).hexdigest()[:16]}", flush=True)
res = np.zeros((len(hs), 3, reps))
for k in range(reps):
    for j, r in enumerate(hs):
        res[j, 0, k] = shadow_ms(r); res[j, 1, k] = frame_ms(r, 0); res[j, 2, k] = frame_ms(r, 1)
m = np.median(res, axis=2)
arrow = lambda col: " -> ".join(f"{m[j, col]:.4f}" for j in range(len(hs)))
print(f"{lib}: shadow pass alone {arrow(0)} ms   whole frame, map redrawn {arrow(1)} ms   static sun {arrow(2)} ms   (modes small:owner {os.environ.get('MODES', '0:-1,1:-1')}, medians of {reps})", flush=True)
for r in hs: r.close()
