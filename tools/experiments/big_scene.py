"""stress: millions of (mostly sub-pixel) triangles through the prepass and the frame paths; checks the frame path against
the G-buffer path and prints the rasteriser's work counts."""
import sys, time, numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..' if os.path.basename(os.path.dirname(os.path.abspath(__file__))) == 'tools' else os.path.join('..', '..')))
import __graft_entry__ as e
pkg = e.load_package()
S = pkg.scenes
for nu, nv, copies in ((1024, 512, 4), (2048, 1024, 2)):
    sc = S.config2(scale=1.0)
    v, i = S.uv_sphere(1.0, nu, nv)
    sc.meshes = [(v, i, 0)] + sc.meshes[1:]
    objs = sc.desc.objects.copy()
    extra = np.repeat(objs[:1], copies - 1)
    for k in range(copies - 1):
        extra[k]["trs"][12] += 2.2 * (k + 1)
    sc.desc.objects = np.concatenate([objs, extra])
    r = sc.upload(pkg.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
    t = time.perf_counter(); img = r.render_frame(sc.desc, sc.settings); dt0 = time.perf_counter() - t
    st = r.stats()
    t = time.perf_counter()
    for _ in range(5): r.render_frame_device(sc.desc, sc.settings, None)
    r.flush(); dt = (time.perf_counter() - t) / 5
    r.set_option("visbuffer", 0)
    ref = r.render_frame(sc.desc, sc.settings)
    print(f"{sc.n_triangles} triangles at {sc.width}x{sc.height}: first frame {dt0*1e3:.1f} ms, then {dt*1e3:.3f} ms per frame; forward {st[0]} records / {st[1]} items, "
          f"shadow {st[2]} / {st[3]}; frame path == G-buffer path: {bool((img == ref).all())}; coverage {(img[..., :3].sum(-1) > 0).mean():.2f}", flush=True)
    r.close()
