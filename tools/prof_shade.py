# minimal program for rocprofv3: build config-3 G-buffer once, then launch k_shade a few times
import sys, numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import __graft_entry__ as e
pkg=e.load_package()
opts=[a for a in sys.argv[1:] if "=" in a]                 # name=value: arctic_set_option before anything is rendered
args=[a for a in sys.argv[1:] if "=" not in a]
mode=args[0] if len(args)>0 else "full"
cfg=int(args[1]) if len(args)>1 else 3      # BASELINE config number (default 3: the metric's)
sc=pkg.scenes.CONFIGS[cfg](scale=1.0)
r=pkg.Renderer(sc.width,sc.height,sc.shadow_size,sc.max_lights)
for o in opts: r.set_option(o.split('=')[0], int(o.split('=')[1]))   # (before the upload: texture_tiling applies to materials created afterwards)
sc.upload(r)
r.pass_shadow_map(sc.desc); r.pass_gbuffer(sc.desc); r.flush()
if mode=="nolights": r.update_lights(sc.lights[:0])
if mode=="nocull": r.set_option("culling",0)
if mode.startswith("dbg"): r.set_option("debug", int(mode[3:])); r.update_lights(sc.lights[:0])
for i in range(5):
    if mode == "frame": (r.pass_shadow_map(sc.desc), r.pass_gbuffer(sc.desc), r.pass_shade(sc.desc, sc.settings))   # every pass, through the G-buffer
    elif mode == "render_frame": r.render_frame_device(sc.desc, sc.settings, None)                                   # what the application calls
    else: r.pass_shade(sc.desc, sc.settings)
r.flush()
r.close()
