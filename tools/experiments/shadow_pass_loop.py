"""30 shadow passes of config 3 (4000^2 map, 232 k triangles) for `rocprofv3 --kernel-trace --stats` (tools/experiments/shadow_raster_ab.sh);
prints a checksum of the map so that builds can be compared bit for bit.  The library is chosen with ARCTIC_HIP_LIBRARY."""
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import __graft_entry__ as e  # noqa: E402

pkg = e.load_package()
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
sc = pkg.scenes.CONFIGS[cfg](scale=1.0)
r = sc.upload(pkg.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
r.set_option("shadow_cache", 0)
for _ in range(10):
    r.pass_shadow_map(sc.desc)
r.flush()
t = time.perf_counter()
for _ in range(30):
    r.pass_shadow_map(sc.desc)
r.flush()
ms = (time.perf_counter() - t) / 30 * 1e3
m = r.read_shadow_map()
print(f"{os.environ.get('ARCTIC_HIP_LIBRARY', 'default')}: shadow pass {ms:.4f} ms back to back, map sha1 {hashlib.sha1(m.tobytes()).hexdigest()[:16]}", flush=True)
r.close()
