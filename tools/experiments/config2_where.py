"""Where does the time of config 2's shading pass (1080p, sun only, 2048^2 textures, 2048^2 shadow map) go?  The pass with its material
textures / its shadow test switched off (ARCTIC_OPT_DEBUG bits 0 / 1: timing only, wrong images), rotating three handles so that the
256 MiB Infinity Cache cannot serve re-reads.   usage: python tools/experiments/config2_where.py [config]"""
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import __graft_entry__ as e
import torch
pkg = e.load_package()
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
sc = pkg.scenes.CONFIGS[cfg](scale=1.0)
hs = []
for _ in range(3):
    r = sc.upload(pkg.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
    r.set_stream(torch.cuda.current_stream().cuda_stream)
    r.pass_shadow_map(sc.desc); r.pass_gbuffer(sc.desc); r.flush()
    hs.append(r)
out = torch.empty((sc.height, sc.width, 4), dtype=torch.uint8, device="cuda")
def timed(n=300):
    fns = [h.prepared_pass_shade(sc.desc, sc.settings) for h in hs]
    for k in range(60): fns[k % 3](out.data_ptr())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(n): fns[k % 3](out.data_ptr())
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
_, mat, _, _ = hs[0].read_gbuffer(want=("material",))
covered = int((mat != 0xFFFFFFFF).sum())
print(f"config {cfg}: {sc.width}x{sc.height}, covered pixels {covered} of {mat.size} ({covered / mat.size:.3f})")
for name, dbg in (("the pass", 0), ("without material textures", 1), ("without the shadow test", 2), ("without either", 3)):
    for h in hs: h.set_option("debug", dbg)
    t = timed()
    print(f"{name:28s} {t:.4f} ms   ({covered * 80 / t / 1e6:.0f} GB/s on the 80 B of the covered pixels, {mat.size * 80 / t / 1e6:.0f} GB/s on every pixel of the target)")
for h in hs: h.set_option("debug", 0)
for T in (1, 2):
    for h in hs: h.set_option("tiles_per_wave", T)
    print(f"tiles per wave {T}: {timed():.4f} ms")
