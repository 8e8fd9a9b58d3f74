"""render one frame with the HIP library and write it as a PNG (no imaging library needed).
usage: python tools/render_scene.py out.png [config 1-5 | path.gltf/.glb] [scale] [--sky]
For a glTF file the camera orbits the scene's bounding box; lights: the default sun (src/app.hpp:51-55)."""
import struct, sys, zlib
import numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..' if os.path.basename(os.path.dirname(os.path.abspath(__file__))) == 'tools' else os.path.join('..', '..')))
import __graft_entry__ as e
pkg = e.load_package()


def write_png(path, rgba):
    h, w = rgba.shape[:2]
    raw = b"".join(b"\0" + rgba[y, :, :3].tobytes() for y in range(h))
    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))
    open(path, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


out = sys.argv[1]
what = sys.argv[2] if len(sys.argv) > 2 else "3"
scale = float(sys.argv[3]) if len(sys.argv) > 3 and not sys.argv[3].startswith("--") else 0.25
sky = "--sky" in sys.argv
if what.isdigit():
    sc = pkg.scenes.CONFIGS[int(what)](scale=scale)
    if sky and sc.environment is None:
        sc.environment = pkg.scenes.synthetic_hdri(1024, 512)
    r = sc.upload(pkg.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
    img = r.render_frame(sc.desc, sc.settings)
else:
    from importlib import import_module
    g = import_module("arctic_renderer_amd.gltf").load(what)
    pts = np.concatenate([v["position"] for v, _, _ in g.meshes])
    lo, hi = pts.min(0), pts.max(0)
    c, rad = (lo + hi) / 2, float(np.linalg.norm(hi - lo)) / 2 + 1e-3
    w, h = 1280, 720
    eye = c + np.array([0.0, 0.3 * rad, 2.2 * rad])
    desc = pkg.scene.SceneDesc(camera=dict(eye=tuple(eye), rotation=(-8.0, -90.0), aspect=w / h, fov_y=45.0, z_near_far=(0.01 * rad, 100.0 * rad)),
                               ambient=0.1, sun=pkg.scenes.DEFAULT_SUN, objects=g.objects)
    r = g.upload(pkg.Renderer(w, h, 4000, 16))
    if sky:
        r.create_hdri(pkg.scenes.synthetic_hdri(1024, 512))
    img = r.render_frame(desc, (2, 2.2, 1.0))
write_png(out, img)
print(f"wrote {out}: {img.shape[1]}x{img.shape[0]}, mean {img[..., :3].mean():.1f}")
