import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import __graft_entry__ as e
pkg=e.load_package()
from oracle import oracle as O
from test_oracle_kat import radiance64, tonemap64, unit
sc=pkg.scenes.CONFIGS[3](scale=0.2)
r=sc.upload(pkg.Renderer(sc.width,sc.height,sc.shadow_size,sc.max_lights)); r.set_option("keep_float_output",1)
o=sc.upload(O.Oracle(sc.width,sc.height,sc.shadow_size,sc.max_lights))
o.pass_shadow_map(sc.desc); o.pass_gbuffer(sc.desc); r.pass_shadow_map(sc.desc); r.pass_gbuffer(sc.desc)
st=(0,2.2,1.0)
o.pass_shade(sc.desc, st); r.pass_shade(sc.desc, st)
oldr,ohdr,_=o.read_output(); hldr,hhdr,_=r.read_output()
err=np.abs(oldr-hldr); 
idx=np.argsort(err.max(-1).ravel())[::-1][:5]
attrs,mat,_,_=o.read_gbuffer()
eye=np.array(sc.desc.camera["eye"],np.float64); sun_dir=O.dir_from_rot(sc.desc.sun["rotation"]).astype(np.float64)
for k in idx:
    y,x=divmod(int(k),sc.width)
    a=attrs[y,x].astype(np.float64); m=int(mat[y,x])
    s=o.fetch_surface(m, float(attrs[y,x,0]), float(attrs[y,x,1]), attrs[y,x,2:11])
    base=s[0:3].astype(np.float64); n=s[6:9].astype(np.float64); metal=float(s[9]); rough=float(s[10])
    n=unit(n)
    world=a[11:14]; wo=unit(eye-world)
    shadow=O.calculate_shadow(o.read_shadow_map(), attrs[y,x,14:18]); lit=1-shadow
    Lo=lit*radiance64(n,wo,-sun_dir,np.array(sc.desc.sun["color"],np.float64),base,metal,rough)
    contrib=[]
    for L in sc.lights:
        d=L["position"].astype(np.float64)-world; dist=np.linalg.norm(d)
        c=lit*radiance64(n,wo,d/dist,L["color"].astype(np.float64)/dist**2,base,metal,rough); Lo=Lo+c; contrib.append(c.max())
    col=Lo+0.1*base
    tm,out=tonemap64(0,col)
    print(f"px ({y},{x}) mat {m} rough {rough:.4f} metal {metal:.3f} lit {lit:.2f} ndwo {n@wo:.4f}")
    print("   exact64 hdr", col, "ldr", out)
    print("   oracle32 hdr", ohdr[y,x], "ldr", oldr[y,x], " err vs exact", np.abs(oldr[y,x]-out).max())
    print("   hip      hdr", hhdr[y,x], "ldr", hldr[y,x], " err vs exact", np.abs(hldr[y,x]-out).max())
    print("   top light contribs", sorted(contrib)[-3:])
