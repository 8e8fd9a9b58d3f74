"""how much frames in flight would buy: 1, 2, 3 independent handles (each with its own streams and tables) render config 3 in turn.
usage: python tools/experiments/frames_in_flight.py"""
import sys, time, numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import __graft_entry__ as e
import torch
pkg = e.load_package()
sc = pkg.scenes.config3(scale=1.0)
hs = [sc.upload(pkg.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights)) for _ in range(3)]
outs = [torch.empty((sc.height, sc.width, 4), dtype=torch.uint8, device="cuda") for _ in hs]
for cache in (1, 0):
    for n in (1, 2, 3):
        use = hs[:n]
        for h in use: h.set_option("shadow_cache", cache)
        for i in range(6):
            for h, o in zip(use, outs): h.render_frame_device(sc.desc, sc.settings, o.data_ptr())
        for h in use: h.flush()
        N = 60
        t = time.perf_counter()
        for i in range(N // n):
            for h, o in zip(use, outs): h.render_frame_device(sc.desc, sc.settings, o.data_ptr())
        for h in use: h.flush()
        dt = (time.perf_counter() - t) / (N // n * n)
        print(f"{'static' if cache else 'moving'} sun, {n} handle(s) in flight: {dt*1e3:.3f} ms per frame", flush=True)
