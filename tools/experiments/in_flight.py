"""whole frames with 1 / 2 / 3 frames in flight (ARCTIC_OPT_FRAMES_IN_FLIGHT), static sun and shadow map redrawn every frame.
usage: python tools/experiments/in_flight.py [config[:scale] ...]"""
import sys, os, time, copy
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import __graft_entry__ as e
TORCH_STREAM = os.environ.get("ARCTIC_USE_TORCH_STREAM") == "1"   # the caller brings its stream (what bench.py does): the handle's own stream is idle
if TORCH_STREAM:
    import torch
    torch.cuda.set_device(0); torch.zeros(1, device="cuda")   # (torch first, as in bench.py)
pkg = e.load_package()
for arg in sys.argv[1:] or ("3", "2", "1"):
    cfg, scale = (int(arg.split(":")[0]), float(arg.split(":")[1])) if ":" in arg else (int(arg), 1.0)
    sc = pkg.scenes.CONFIGS[cfg](scale=scale)
    r = sc.upload(pkg.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
    if TORCH_STREAM:
        r.set_stream(torch.cuda.current_stream().cuda_stream)
    moving = [copy.deepcopy(sc.desc) for _ in range(2)]
    moving[1].sun = dict(moving[1].sun, rotation=(moving[1].sun["rotation"][0] - 1.5, moving[1].sun["rotation"][1] + 4.0))
    def timed(fn, n):
        for k in range(20): fn(k)
        r.flush(); t = time.perf_counter()
        for k in range(n): fn(k)
        r.flush()
        return (time.perf_counter() - t) / n * 1e3
    t0 = time.time()
    while time.time() - t0 < 0.4: r.render_frame_device(sc.desc, sc.settings, None)
    r.flush()
    for rep in range(2):
        for n in (2, 3, 1):
            r.set_option("frames_in_flight", n)
            f = timed(lambda k: r.render_frame_device(sc.desc, sc.settings, None), 200)
            m = timed(lambda k: r.render_frame_device(moving[k & 1], sc.settings, None), 200)
            print(f"config {cfg} {sc.width}x{sc.height}, {n} in flight: frame static {f:.4f} ms  moving sun {m:.4f} ms", flush=True)
    r.close()
