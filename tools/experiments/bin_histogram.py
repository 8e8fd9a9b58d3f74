"""work items per 16x16 block (forward and shadow prepass) of a configuration at full size: how full the owners' bins get"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import __graft_entry__ as e
pkg = e.load_package()
for cfg in [int(a) for a in sys.argv[1:]] or (3, 2, 5):
    sc = pkg.scenes.CONFIGS[cfg](scale=1.0)
    r = sc.upload(pkg.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
    r.set_option("raster_owner", 3); r.set_option("debug", 512)
    r.pass_shadow_map(sc.desc); r.pass_gbuffer(sc.desc); r.flush()
    st = r.stats()
    for name, c, items in (("forward", r.bin_counts(False), int(st[1])), ("shadow", r.bin_counts(True), int(st[3]))):
        c = c.ravel().astype(np.int64)
        q = np.percentile(c, [50, 90, 99, 99.9])
        print(f"config {cfg} {name}: {c.size} blocks, {items} items ({c.sum()} exact), mean {c.mean():.2f}, p50/p90/p99/p99.9 {q}, max {c.max()}; empty {np.mean(c == 0):.3f}; "
              + "; ".join(f"over {k}: {np.maximum(c - k, 0).sum() / max(c.sum(), 1):.4f} of the items in {np.mean(c > k):.4f} of the blocks" for k in (8, 16, 32, 64)), flush=True)
    r.close()
