"""ARCTIC_OPT_SMALL_TRIANGLES off / on over random triangle soups (the generator of tests/test_gpu_small_triangles.py with more seeds, sizes and suns): the shadow maps must be
identical, bit for bit.   usage: python tools/experiments/small_fuzz.py [cases]"""
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import __graft_entry__ as e
pkg = e.load_package()
from test_gpu_small_triangles import soup, shadow
hip = pkg.renderer
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = 0
for case in range(cases):
    rng = np.random.default_rng(9000 + case)
    S = int(rng.choice([64, 97, 256, 511, 1000, 2048, 4000]))
    n = int(rng.integers(2000, 40000))
    lo = float(np.exp(rng.uniform(np.log(0.002), np.log(0.05)))); hi = float(np.exp(rng.uniform(np.log(0.1), np.log(8.0))))
    mesh = soup(pkg, rng, n, lo, hi, float(rng.uniform(10, 30)))
    sun = dict(position=tuple(float(x) for x in rng.uniform((-15, 20, -15), (15, 40, 15))), rotation=(float(rng.uniform(-89, -30)), float(rng.uniform(-180, 180))), color=(8.0, 8.0, 8.0))
    desc = pkg.scene.SceneDesc(camera=dict(eye=(0, 5, 0), rotation=(-15.0, 0.0), aspect=2.0, fov_y=45.0, z_near_far=(0.1, 100.0)), ambient=0.1, sun=sun,
                               objects=pkg.scene.make_objects([(np.eye(4, dtype=np.float32), 0)]))
    sc = pkg.scenes.SyntheticScene("soup", 64, 32, S, 16, [pkg.scenes.fallback_textures()], [mesh + (0,)], desc, np.zeros(0, pkg.scene.LIGHT_DTYPE), (0, 2.2, 1.0))
    try:
        (m0, s0), (m1, s1) = shadow(hip, sc, 0), shadow(hip, sc, 1)
    except Exception as ex:   # (a soup of large triangles can overflow the default item table: not this tool's subject)
        print(f"case {case}: S {S}, {n} triangles {lo:.4f}..{hi:.2f} m: skipped ({str(ex)[:60]})", flush=True); continue
    same = bool(np.array_equal(m0, m1)) and s0[2] == s1[2]
    bad += 0 if same else 1
    print(f"case {case}: S {S}, {n} triangles {lo:.4f}..{hi:.2f} m, sun {sun['rotation'][0]:.0f}/{sun['rotation'][1]:.0f}: records {s1[2]}, items {s0[3]} -> {s1[3]}, drawn texels {(m0 != 0x3F800000).mean():.4f}: {'same' if same else 'DIFFERENT'}", flush=True)
print(f"{cases} cases, {bad} failures")
sys.exit(1 if bad else 0)
