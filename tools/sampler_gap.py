"""How much of the image depends on what the reference leaves to the D3D12 sampler hardware?  CPU only (the oracle; test infrastructure).

The reference names its sampler (MIN_MAG_MIP_LINEAR + WRAP, forward_pass.cpp:38-51) and its texture formats (renderer.cpp:483-548) and
leaves the filter arithmetic to the GPU.  The oracle -- and the HIP kernels, which match it -- filter with full-fp32 weights and
decode sRGB per texel before filtering.  This tool renders BASELINE configs 1-3 at test scale with the oracle's sampler VARIANTS
(oracle/arctic_oracle.cpp SAMPLER_*: 8-bit filter weights for the material textures, sRGB decoded after filtering, 8-bit weights for
the shadow map's PCF taps) and reports, per variant, the distance to the default: max / 99.99th percentile / mean of |LDR delta| over
all channels, the share of RGBA8 channels that change and by how many steps.  The unpinned-parity gap is bounded by these numbers
instead of being asserted (README).

usage: python tools/sampler_gap.py [out.json]          (writes profiles/r4_sampler_gap.json by default)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

VARIANTS = {"q8_material_weights": 1, "srgb_decode_after_filter": 2, "q8_shadow_weights": 4, "all_three": 7}
CASES = [("config1", 0.25), ("config2", 0.15), ("config3", 0.1)]        # the test scales of tests/test_gpu_parity.py


def render(O, sc, mode, threads):
    o = sc.upload(O.Oracle(sc.width, sc.height, sc.shadow_size, sc.max_lights))
    o.set_sampler_mode(mode)
    o.render_frame(sc.desc, sc.settings, threads=threads)
    ldr, _, rgba = o.read_output()
    o.close()
    return ldr.copy(), rgba.copy()


def measure(cases=CASES, threads=None):
    from oracle import oracle as O
    O.build()
    pkg = entry.load_package()
    threads = threads or min(8, O.hardware_threads() or 8)
    out = {"what": "oracle (float64 BRDF, default sampler) against the same oracle with a sampler variant; |LDR delta| over all pixels and channels",
           "variants": {"q8_material_weights": "texel coordinate snapped to 1/256 texel before index / weight split (material textures)",
                        "srgb_decode_after_filter": "sRGB texels filtered as stored, the result decoded",
                        "q8_shadow_weights": "the same 8-bit weights for the 25 PCF taps of the shadow map",
                        "all_three": "all of the above"},
           "cases": {}}
    for name, scale in cases:
        sc = getattr(pkg.scenes, name)(scale=scale)
        base_ldr, base_rgba = render(O, sc, 0, threads)
        row = {"size": [sc.width, sc.height], "scale": scale, "shadow_size": sc.shadow_size, "point_lights": len(sc.lights)}
        for vname, mode in VARIANTS.items():
            if mode == 4 and not sc.shadow_size:
                continue
            ldr, rgba = render(O, sc, mode, threads)
            d = np.abs(ldr.astype(np.float64) - base_ldr.astype(np.float64)).ravel()
            s = np.abs(rgba[..., :3].astype(np.int16) - base_rgba[..., :3].astype(np.int16))
            row[vname] = {"max": float(d.max()), "p9999": float(np.quantile(d, 0.9999)), "p99": float(np.quantile(d, 0.99)), "mean": float(d.mean()),
                          "share_above_1e-4": float((d > 1e-4).mean()), "rgba8_mismatch_rate": float((s != 0).mean()), "rgba8_max_steps": int(s.max())}
        out["cases"][f"{name}@{scale}"] = row
    return out


if __name__ == "__main__":
    t = time.time()
    res = measure()
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r4_sampler_gap.json")
    json.dump(res, open(path, "w"), indent=1)
    for case, row in res["cases"].items():
        for v in VARIANTS:
            if v in row:
                m = row[v]
                print(f"{case:14s} {v:26s} max {m['max']:.2e}  p99.99 {m['p9999']:.2e}  mean {m['mean']:.2e}  > 1e-4: {m['share_above_1e-4']:.4f}  "
                      f"RGBA8 channels changed {m['rgba8_mismatch_rate']:.4f} (up to {m['rgba8_max_steps']} steps)")
    print(f"[{time.time() - t:.0f} s] -> {path}")
