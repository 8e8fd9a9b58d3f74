#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz from the CPU oracle (float64 BRDF arithmetic).

    python tests/golden/make_golden.py

The reference ships no golden images for this path, so these fixtures pin the
ORACLE against regressions -- they are inputs (seeded scene parameters) and expected outputs (float LDR +
RGBA8 + G-buffer checksums), not reference source.  Cases follow SURVEY.md 8(c): 64x64 / 128x128 renders of
the seeded scenes, per tonemapper, with and without shadow map, with 0 / 1 / 16 lights.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402
from oracle import oracle as O  # noqa: E402

CASES = [
    # name, config, scale, override size, shadow, n_lights, tonemapper
    ("c1_64_reinhard", 1, 0.125, 64, False, 0, 0),
    ("c2_128_reinhard_shadow", 2, 0.07, None, True, 0, 0),
    ("c2_128_exposure_noshadow", 2, 0.07, None, False, 0, 1),
    ("c3_128_aces_16lights", 3, 0.035, None, True, 16, 2),
    ("c3_128_reinhard_1light", 3, 0.035, None, True, 1, 0),
    ("c3_64_aces_noshadow_16lights", 3, 0.02, None, False, 16, 2),
]


def build(case):
    pkg = entry.load_package()
    name, cfg, scale, size, shadow, n_lights, tm = case
    sc = pkg.scenes.CONFIGS[cfg](scale=scale, tex=64)
    shadow_size = sc.shadow_size if shadow else 0
    lights = sc.lights[:n_lights] if len(sc.lights) >= n_lights else pkg.scenes.random_lights(
        np.random.default_rng(7), n_lights, (-3, 0.5, -3), (3, 3, 3))
    o = O.Oracle(sc.width, sc.height, shadow_size, 16)
    for d, n, m in sc.materials:
        o.create_material(d, n, m)
    for v, i, mat in sc.meshes:
        o.create_mesh(v, i, mat)
    o.update_lights(lights)
    settings = (tm, 2.2, 0.8 if tm == 1 else 1.0)
    return sc, o, settings


def render(case):
    sc, o, settings = build(case)
    rgba = o.render_frame(sc.desc, settings, threads=4)
    ldr, hdr, _ = o.read_output()
    attrs, mat, depth, tri = o.read_gbuffer()
    return dict(rgba8=rgba, ldr=ldr.astype(np.float32),
                gbuffer_sha=np.frombuffer(hashlib.sha256(attrs.tobytes() + mat.tobytes() + depth.tobytes() + tri.tobytes()).digest(), np.uint8),
                coverage=np.float64((mat != 0xFFFFFFFF).mean()))


if __name__ == "__main__":
    O.build()
    for case in CASES:
        out = render(case)
        np.savez_compressed(os.path.join(HERE, case[0] + ".npz"), **out)
        print(case[0], out["rgba8"].shape, "coverage %.3f" % out["coverage"])
