import sys, time, numpy as np
sys.path.insert(0,'/root/repo')
import __graft_entry__ as e
import torch
pkg=e.load_package()
for cfg in (3,2,1):
    sc=pkg.scenes.CONFIGS[cfg](scale=1.0)
    r=sc.upload(pkg.Renderer(sc.width,sc.height,sc.shadow_size,sc.max_lights))
    out=torch.empty((sc.height,sc.width,4),dtype=torch.uint8,device="cuda")
    for i in range(3): r.render_frame_device(sc.desc, sc.settings, out.data_ptr())
    r.flush(); t=time.perf_counter(); N=20
    for i in range(N): r.render_frame_device(sc.desc, sc.settings, out.data_ptr())
    r.flush(); dt=(time.perf_counter()-t)/N
    print(f"config {cfg}: {sc.width}x{sc.height} {sc.n_triangles} tris, whole frame (shadow raster + prepass + shading) {dt*1e3:.3f} ms = {1/dt:.0f} fps, {sc.width*sc.height/dt/1e6:.0f} Mpx/s", flush=True)
    ms=r.time_shade(sc.desc, sc.settings, warmup=3, iters=20); print(f"    shading pass alone {np.mean(ms):.4f} ms")
    r.close()
