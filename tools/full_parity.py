import sys, time, numpy as np
sys.path.insert(0,'/root/repo')
import __graft_entry__ as e
pkg=e.load_package()
from oracle import oracle as O
cfg=int(sys.argv[1]) if len(sys.argv)>1 else 3
scale=float(sys.argv[2]) if len(sys.argv)>2 else 1.0
sc=pkg.scenes.CONFIGS[cfg](scale=scale)
r=sc.upload(pkg.Renderer(sc.width,sc.height,sc.shadow_size,sc.max_lights)); r.set_option("keep_float_output",1)
o=sc.upload(O.Oracle(sc.width,sc.height,sc.shadow_size,sc.max_lights))
t=time.time(); o.pass_shadow_map(sc.desc); o.pass_gbuffer(sc.desc); print("oracle geom",time.time()-t, flush=True)
t=time.time(); r.pass_shadow_map(sc.desc); r.pass_gbuffer(sc.desc); r.flush(); print("hip geom",time.time()-t, r.stats(), o.stats(), flush=True)
if sc.shadow_size:
    a,b=o.read_shadow_map(), r.read_shadow_map()
    print("shadow equal", np.array_equal(a.view(np.uint32),b.view(np.uint32)), (a<1).mean(), (b<1).mean(), "ndiff", (a.view(np.uint32)!=b.view(np.uint32)).sum())
oa,om,od,ot=o.read_gbuffer(); ha,hm,hd,ht=r.read_gbuffer()
print("mat eq",np.array_equal(om,hm),"tri eq",np.array_equal(ot,ht),"depth eq",np.array_equal(od.view(np.uint32),hd.view(np.uint32)),"attr eq",np.array_equal(oa.view(np.uint32),ha.view(np.uint32)), "ndiff attr", (oa.view(np.uint32)!=ha.view(np.uint32)).sum(), flush=True)
t=time.time(); o.pass_shade(sc.desc, sc.settings, threads=O.hardware_threads()); print("oracle shade", time.time()-t, flush=True)
r.pass_shade(sc.desc, sc.settings); 
oldr,ohdr,orgba=o.read_output(); hldr,hhdr,hrgba=r.read_output()
err=np.abs(oldr-hldr); print("max err", err.max(), "p99.99", np.quantile(err,0.9999), "mean", err.mean(), "rgba8 mismatch", (orgba!=hrgba).mean(), "max", np.abs(orgba.astype(int)-hrgba.astype(int)).max())
lit=(ohdr.sum(-1) > (0.1*1.01)).mean(); print("hdr mean", ohdr.mean(), "frac hdr>0.3:", (ohdr.max(-1)>0.3).mean())
from PIL import Image
Image.fromarray(hrgba[...,:3]).resize((sc.width//4, sc.height//4)).save('/root/repo/gpurun_out/full_c%d.png'%cfg)
