import sys, numpy as np, torch
sys.path.insert(0,'/root/repo')
import __graft_entry__ as e
pkg=e.load_package()
from importlib import import_module
sh=import_module("arctic_renderer_amd.sharding")
sc=pkg.scenes.CONFIGS[3](scale=float(sys.argv[1]) if len(sys.argv)>1 else 1.0)
full=sc.upload(pkg.Renderer(sc.width,sc.height,sc.shadow_size,sc.max_lights)); ref=full.render_frame(sc.desc, sc.settings); full.close()
world=4
for rank in range(world):
    r=sc.upload(pkg.Renderer(sc.width,sc.height,sc.shadow_size,sc.max_lights,band_rows=16,shard=(rank,world)))
    rows=sh.owned_rows(sc.height,rank,world,16)
    img=r.render_frame(sc.desc, sc.settings)
    ok1=np.array_equal(img, ref[rows])
    r.pass_shadow_map(sc.desc); r.pass_gbuffer(sc.desc)
    out=torch.empty((r.rows,sc.width,4),dtype=torch.uint8,device="cuda")
    r.set_stream(torch.cuda.current_stream().cuda_stream)
    shade=r.prepared_pass_shade(sc.desc, sc.settings)
    for i in range(3): shade(out.data_ptr())
    torch.cuda.synchronize()
    o=out.cpu().numpy()
    ok2=np.array_equal(o, ref[rows])
    bad=np.nonzero((o!=ref[rows]).any(axis=(1,2)))[0]
    print(rank, "render_frame ok", ok1, "pass_shade(d_out) ok", ok2, "bad rows", len(bad), bad[:10], "of", len(rows), flush=True)
    r.close()
