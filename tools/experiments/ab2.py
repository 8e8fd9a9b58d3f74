"""A/B of builds of the library on ONE box, alternating, with the clocks warmed up: the shading pass of config 3 (4K) at a few light
counts, automatic light loop, medians over the alternations.   usage: python tools/experiments/ab2.py libA.so libB.so [reps]
(worker mode: ab2.py --worker: prints one line of times for the library in ARCTIC_HIP_LIBRARY)"""
import os, sys, subprocess, numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
COUNTS = (64, 16, 0)
NAMES = ["64 lights", "16 lights", "0 lights", "frame", "frame + shadow"]
if len(sys.argv) > 1 and sys.argv[1] == "--worker":
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    import __graft_entry__ as e
    pkg = e.load_package()
    sc = pkg.scenes.CONFIGS[4](scale=1.0)
    r = sc.upload(pkg.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
    for opt in sys.argv[2:]:
        r.set_option(opt.split("=")[0], int(opt.split("=")[1]))
    r.pass_shadow_map(sc.desc); r.pass_gbuffer(sc.desc); r.flush()
    r.update_lights(sc.lights[:64])
    r.time_shade(sc.desc, sc.settings, warmup=300, iters=10)   # ~70 ms of load: the clocks settle
    out = []
    for n in COUNTS:
        r.update_lights(sc.lights[:n])
        out.append(float(np.median(r.time_shade(sc.desc, sc.settings, warmup=20, iters=60))))
    # whole frames (static sun / shadow map redrawn every frame), 64 lights, 40 frames enqueued back to back
    import time
    r.update_lights(sc.lights[:64])
    for cache in (1, 0):
        r.set_option("shadow_cache", cache)
        for i in range(10): r.render_frame_device(sc.desc, sc.settings, None)
        r.flush(); t = time.perf_counter()
        for i in range(40): r.render_frame_device(sc.desc, sc.settings, None)
        r.flush(); out.append((time.perf_counter() - t) / 40 * 1e3)
    print("TIMES " + " ".join(f"{t:.4f}" for t in out), flush=True)
    sys.exit(0)
specs = [a for a in sys.argv[1:] if not a.isdigit()]
reps = int(next((a for a in sys.argv[1:] if a.isdigit()), 3))
res = {s: [] for s in specs}
for _ in range(reps):
    for s in specs:   # a spec = path to a library, optionally followed by ,option=value,...
        lib, *opts = s.split(",")
        env = dict(os.environ, ARCTIC_HIP_LIBRARY=os.path.abspath(lib), ARCTIC_HIP_LIBRARY_OLDER="1")   # (an earlier round's build may lack this round's entry points)
        o = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker"] + opts, env=env, capture_output=True, text=True).stdout
        line = [l for l in o.splitlines() if l.startswith("TIMES")]
        if line: res[s].append([float(x) for x in line[0].split()[1:]])
        else: print(f"{s}: the worker printed no times:\n{o[-2000:]}", flush=True)
for s in specs:
    if not res[s]: continue
    a = np.array(res[s])
    print(f"{s}: " + "  ".join(f"{n} {np.median(a[:, i]):.4f} ms ({a[:, i].min():.4f}-{a[:, i].max():.4f})" for i, n in enumerate(NAMES)), flush=True)
