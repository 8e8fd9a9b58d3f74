"""whole frames of a configuration for a rocprofv3 kernel trace: static sun, then the shadow map redrawn every frame.
usage: rocprofv3 --kernel-trace --stats ... -- python3 tools/experiments/frames_for_trace.py [config] [frames] [option=value ...]
SHARD=band_rows,index,count in the environment: the frames of one interleaved shard."""
import sys, os, copy
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import __graft_entry__ as e
pkg = e.load_package()
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
sc = pkg.scenes.CONFIGS[cfg](scale=1.0)
shard = [int(x) for x in os.environ["SHARD"].split(",")] if os.environ.get("SHARD") else None
r = sc.upload(pkg.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights, **(dict(band_rows=shard[0], shard=(shard[1], shard[2])) if shard else {})))
for opt in sys.argv[3:]:
    r.set_option(opt.split("=")[0], int(opt.split("=")[1]))
for cache in (1, 0):
    r.set_option("shadow_cache", cache)
    for i in range(n):
        r.render_frame_device(sc.desc, sc.settings, None)
    r.flush()
print('stats', r.stats())
r.close()
