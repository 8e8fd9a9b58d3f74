"""whole-frame and shading-pass times of the BASELINE configurations at full size (secondary figures of SURVEY 8d).
usage: python tools/frame_time.py [config ...]   (default 3 2 1; 4 and 5 are the 256- and 1024-light cases)"""
import sys, time, numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import __graft_entry__ as e
import torch
pkg = e.load_package()
for cfg in [int(a) for a in sys.argv[1:]] or (3, 2, 1):
    sc = pkg.scenes.CONFIGS[cfg](scale=1.0)
    envs = (None, sc.environment if sc.environment is not None else pkg.scenes.synthetic_hdri(2048, 1024)) if cfg in (1, 2, 5) else (None,)
    for env in envs:
        sc.environment = env
        r = sc.upload(pkg.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
        out = torch.empty((sc.height, sc.width, 4), dtype=torch.uint8, device="cuda")
        N = 20 if cfg < 5 else 5
        for cache in (1, 0):   # 1: the shadow map is redrawn only when the sun / objects change (default); 0: every frame, like the reference
            r.set_option("shadow_cache", cache)
            for i in range(3): r.render_frame_device(sc.desc, sc.settings, out.data_ptr())
            r.flush(); t = time.perf_counter()
            for i in range(N): r.render_frame_device(sc.desc, sc.settings, out.data_ptr())
            r.flush(); dt = (time.perf_counter() - t) / N
            print(f"config {cfg}{' + environment map' if env is not None else ''}: {sc.width}x{sc.height} {sc.n_triangles} tris, {len(sc.lights)} lights, "
                  f"whole frame ({'static sun: visibility prepass + shading' if cache else 'shadow raster + visibility prepass + shading'}) "
                  f"{dt*1e3:.3f} ms = {1/dt:.0f} fps, {sc.width*sc.height/dt/1e6:.0f} Mpx/s", flush=True)
        ms = r.time_shade(sc.desc, sc.settings, warmup=3, iters=N)
        r.set_option("count_light_evals", 1); r.pass_shade(sc.desc, sc.settings); r.flush(); st = r.stats(); r.set_option("count_light_evals", 0)
        _, mat, _, _ = r.read_gbuffer(want=("material",)); cov = int((mat != 0xFFFFFFFF).sum())
        print(f"    shading pass {np.mean(ms):.4f} ms; covered {cov/mat.size:.3f}, lit {int(st[6])/max(cov,1):.3f} of covered, "
              f"{int(st[5])/1e6:.1f} M light evaluations; {cov*80/np.mean(ms)/1e6:.0f} GB/s algorithmic", flush=True)
        r.close()
