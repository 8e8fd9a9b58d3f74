"""shading pass and whole frames of the small configurations with T = 1 / 2 / 3 tiles per wave (ARCTIC_OPT_TILES_PER_WAVE): a 1080p frame is
16 k waves at T = 2, fewer than the chip has wave slots.   usage: python tools/experiments/small_frames.py [config[:scale] ...]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import __graft_entry__ as e
pkg = e.load_package()
for arg in sys.argv[1:] or ("2", "1"):   # config or config:scale
    cfg, scale = (int(arg.split(":")[0]), float(arg.split(":")[1])) if ":" in arg else (int(arg), 1.0)
    sc = pkg.scenes.CONFIGS[cfg](scale=scale)
    r = sc.upload(pkg.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
    r.pass_shadow_map(sc.desc); r.pass_gbuffer(sc.desc); r.flush()
    r.time_shade(sc.desc, sc.settings, warmup=300, iters=10)
    for rep in range(2):
        for T in (2, 1, 3):
            r.set_option("tiles_per_wave", T)
            ms = float(np.median(r.time_shade(sc.desc, sc.settings, warmup=20, iters=100)))
            for i in range(10): r.render_frame_device(sc.desc, sc.settings, None)
            r.flush(); t = time.perf_counter()
            for i in range(100): r.render_frame_device(sc.desc, sc.settings, None)
            r.flush(); f = (time.perf_counter() - t) / 100 * 1e3
            print(f"config {cfg} {sc.width}x{sc.height} T={T}: shading pass {ms:.4f} ms, whole frame {f:.4f} ms", flush=True)
    r.close()
