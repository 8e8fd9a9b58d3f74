"""Where the time of the shading pass goes, tile by tile (ARCTIC_OPT_TILE_TRACE): per-SIMD timelines of config 3 at 4K.
For a few light counts: the span of the launch, how evenly the SIMDs finish (the tail), the number of resident waves over time,
and the duration of a tile by kind (fully shadowed / with lit pixels) -- what a wave spends on a tile while it shares its
SIMD with six others.   usage: python tools/experiments/tile_trace.py [light counts...]"""
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import __graft_entry__ as e
pkg = e.load_package()
sc = pkg.scenes.CONFIGS[4](scale=1.0)
r = sc.upload(pkg.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
r.pass_shadow_map(sc.desc); r.pass_gbuffer(sc.desc); r.flush()
r.set_option("tile_trace", 1)
for opt in [a for a in sys.argv[1:] if "=" in a]:
    r.set_option(opt.split("=")[0], int(opt.split("=")[1]))
for n in [int(a) for a in sys.argv[1:] if "=" not in a] or (64, 16, 0):
    r.update_lights(sc.lights[:n])
    for _ in range(3): r.pass_shade(sc.desc, sc.settings)
    t = r.tile_trace().reshape(-1, 4)
    t0, t1, hw = t[:, 0].astype(np.int64), t[:, 1].astype(np.int64), t[:, 2]
    core = (t[:, 3] >> 8).astype(np.int64); fast = (t[:, 3] & 1).astype(bool)
    t1 -= t0.min(); t0 -= t0.min()
    span = t1.max()
    dur = np.maximum(t1 - t0, 1)
    hwid = (hw & 0xFFFFFFFF).astype(np.int64); xcc = ((hw >> 32) & 0xF).astype(np.int64); pro = (hw >> 40).astype(np.int64)
    simd = (hwid >> 4) & 3; cu = (hwid >> 8) & 0xF; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 7
    key = ((xcc * 8 + se) * 2 + sh) * 64 + cu * 4 + simd
    uniq, inv = np.unique(key, return_inverse=True)
    last = np.zeros(len(uniq), np.int64); np.maximum.at(last, inv, t1)
    busy = np.zeros(len(uniq), np.int64); np.add.at(busy, inv, dur)
    lit = core > 3 * np.median(core)
    print(f"== {n} point lights: {len(t)} tiles on {len(uniq)} SIMDs in {len(np.unique(xcc))} XCDs; launch span {span * 10} ns")
    print(f"   SIMDs finish at {(np.percentile(last, [0, 1, 10, 50, 90, 100]) * 10).astype(int)} ns (0/1/10/50/90/100 %)")
    print(f"   per XCD: last tile ends at {[int(t1[xcc == x].max()) * 10 for x in np.unique(xcc)]} ns")
    print(f"   tiles per SIMD {np.bincount(inv).min()}..{np.bincount(inv).max()}; resident waves per SIMD (wave time / span): mean {busy.mean() / span:.2f}, min {busy.min() / span:.2f}, max {busy.max() / span:.2f}")
    print(f"   shader-clock ticks per tile: median {int(np.median(core))}; long tiles (> 3 x median, {lit.mean() * 100:.1f} %): median {int(np.median(core[lit])) if lit.any() else 0}; "
          f"fast-tile share {fast.mean() * 100:.1f} %, general tiles: median {int(np.median(core[~fast])) if (~fast).any() else 0}; ticks per 10 ns: {np.median(core[lit] / dur[lit]) if lit.any() else np.median(core / dur):.1f}")
    pro = pro[pro > 0] if (pro > 0).any() else pro      # (recorded for a wave's first tile only)
    print(f"   from kernel entry to the start of a wave's first tile (kernel arguments, order entry): median {int(np.median(pro)) * 10} ns, 90 % {int(np.percentile(pro, 90)) * 10} ns; "
          f"tile's work: median {int(np.median(dur)) * 10} ns, long tiles {int(np.median(dur[lit])) * 10 if lit.any() else 0} ns; wave slot ids in use: {np.bincount(hwid & 15).tolist()}")
    edges = np.linspace(0, span, 21)
    occ = lambda m: " ".join(f"{np.clip(np.minimum(t1[m], b) - np.maximum(t0[m], a), 0, None).sum() / (b - a) / len(uniq):.1f}" for a, b in zip(edges[:-1], edges[1:]))
    print("   resident waves per SIMD over 20 slices of the span: " + occ(np.ones(len(t), bool)))
    print("   ... of which long tiles:                           " + occ(lit), flush=True)
    ty, tx = np.divmod(np.arange(len(t)), r.tile_trace().shape[1])
    rows = [f"{lit[(ty >= a) & (ty < a + 27)].mean() * 100:.0f}" for a in range(0, 270, 27)]
    print("   long-tile share by tenth of the frame, top to bottom (%): " + " ".join(rows), flush=True)
