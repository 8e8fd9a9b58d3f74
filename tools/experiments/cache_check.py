"""is the single-G-buffer timing served from the 256 MiB Infinity Cache?  Two handles with their own G-buffers on ONE
stream (so nothing overlaps): alternating between them doubles the working set; the per-pass time should not change."""
import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..' if os.path.basename(os.path.dirname(os.path.abspath(__file__))) == 'tools' else os.path.join('..', '..')))
import __graft_entry__ as e
pkg = e.load_package()
sc = pkg.scenes.CONFIGS[3](scale=1.0)
hs = []
for i in range(2):
    r = sc.upload(pkg.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
    r.pass_shadow_map(sc.desc); r.pass_gbuffer(sc.desc); r.flush()
    r.set_stream(torch.cuda.current_stream().cuda_stream)
    hs.append(r)
def run(handles, n):
    shade = [h.prepared_pass_shade(sc.desc, sc.settings) for h in handles]
    for s in shade: s(None)
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(n):
        for s in shade: s(None)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / (n * len(handles)) * 1e3
for rep in range(3):
    print(f"one G-buffer  : {run(hs[:1], 200):.4f} ms per pass", flush=True)
    print(f"two G-buffers : {run(hs, 100):.4f} ms per pass (same stream, alternating)", flush=True)
