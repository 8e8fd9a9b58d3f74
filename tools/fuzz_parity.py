"""randomised parity sweep (GPU): HIP path against the CPU oracle over random cameras, suns, settings and scenes.
Bars as in tests/test_gpu_parity.py: shadow map / G-buffer bit-exact, float LDR image within 1e-4 -- except on pixels
where the reference formula itself is ill-conditioned in fp32 (tests/test_oracle_noise_floor.py: grazing views with
n.wo ~ 1e-6, low-roughness highlights): a pixel whose LITERAL fp32 evaluation (oracle precision 32) is itself more than
5e-5 away from the float64 value, or whose float64 value moves by more than 2.5e-5 when tangent frame and world position
move by one fp32 ulp of their vectors' magnitudes, is reported; since the end of round 5 it must hold the 1e-4 like every other pixel unless ARCTIC_FUZZ_ALLOW_ILL=1
(then: within 4x the larger of those two distances instead).
usage: python tools/fuzz_parity.py [n_cases] [seed] [only] [size_factor] [jitter]   (size_factor 8: frames of 2-3 Mpx, where the
rasteriser merges chunks of 32 work items per wave; the default small frames give every wave a single item.  jitter 1: odd frame
and shadow-map sizes -- ragged tiles, scissored windows, a bounds table whose last blocks are cut --, object transforms, materials with unequal image sizes, and the culling / light-loop options of the HIP side)"""
import copy, sys, time
import numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..' if os.path.basename(os.path.dirname(os.path.abspath(__file__))) == 'tools' else os.path.join('..', '..')))
import __graft_entry__ as e
pkg = e.load_package()
from oracle import oracle as O

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
size_factor = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
jitter = len(sys.argv) > 5 and sys.argv[5] == "1"
only = int(sys.argv[3]) if len(sys.argv) > 3 else -1   # run just this case (the random stream is advanced identically) and dump its worst pixel
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2024)
threads = O.hardware_threads()
worst = dict(ldr=0.0, rgba=0.0)
bad = 0
for case in range(n_cases):
    cfg = int(rng.choice([2, 3, 3, 4]))
    scale = min(1.0, {2: 0.2, 3: 0.08, 4: 0.06}[cfg] * size_factor)
    sc = pkg.scenes.CONFIGS[cfg](scale=scale)
    desc = copy.deepcopy(sc.desc)
    if jitter:   # drawn before anything else so that the remaining stream does not depend on the scene
        sc.width = max(9, int(sc.width * rng.uniform(0.8, 1.0)) | 1); sc.height = max(9, int(sc.height * rng.uniform(0.8, 1.0)) | 1)
        if sc.shadow_size: sc.shadow_size = int(sc.shadow_size * rng.uniform(0.6, 1.2)) | int(rng.integers(0, 2))
        desc.camera["aspect"] = sc.width / sc.height
        for _ in range(int(rng.integers(0, 3))):   # a few materials whose three images differ in size: the plain RGBA8 path, one material at a time
            mi = int(rng.integers(0, len(sc.materials)))
            d_, n_, m_ = sc.materials[mi]
            sc.materials[mi] = (d_, np.ascontiguousarray(n_[::2, ::3]) if n_.shape[0] > 4 and n_.shape[1] > 6 else n_, np.ascontiguousarray(m_[::3, ::2]) if m_.shape[0] > 6 and m_.shape[1] > 4 else m_)
        hip_options = dict(culling=int(rng.integers(0, 2)), light_path=int(rng.integers(0, 3)))   # exact culling on / off; auto, scalar or packed light loop
        hip_options["raster_owner"] = (-1, 3, 1, 0)[case % 4]   # the library's choice, block owners in both prepasses, forward only, atomics only (no draw: the stream stays what it was)
        for ob in desc.objects[: 1 + int(rng.integers(0, 3))]:   # move / scale the first few objects (glm column-major trs)
            ob["trs"][12] += float(rng.uniform(-0.3, 0.3)); ob["trs"][13] += float(rng.uniform(0.0, 0.2)); ob["trs"][0] *= float(rng.uniform(0.9, 1.1))
    if cfg == 2:
        az, el, dist = rng.uniform(0, 360), rng.uniform(-5, 40), rng.uniform(2.5, 7.0)
        eye = np.array([dist * np.cos(np.deg2rad(el)) * np.cos(np.deg2rad(az)), 0.8 + dist * np.sin(np.deg2rad(el)), dist * np.cos(np.deg2rad(el)) * np.sin(np.deg2rad(az))])
        desc.camera["eye"] = tuple(float(v) for v in eye)
        desc.camera["rotation"] = (float(-el + rng.uniform(-8, 8)), float(az + 180 + rng.uniform(-10, 10)))
    else:   # inside the atrium (30 x 12 x 14 m): near-plane clipping, grazing walls
        desc.camera["eye"] = (float(rng.uniform(-13, 13)), float(rng.uniform(0.5, 10.5)), float(rng.uniform(-5.5, 5.5)))
        desc.camera["rotation"] = (float(rng.uniform(-60, 60)), float(rng.uniform(0, 360)))
    desc.camera["fov_y"] = float(rng.uniform(30, 90))
    desc.sun["rotation"] = (float(rng.uniform(-89, -25)), float(rng.uniform(0, 360)))
    settings = (int(rng.integers(0, 3)), float(rng.uniform(1.8, 2.6)), float(rng.uniform(0.3, 2.0)))
    env_seed = int(rng.integers(1 << 30)) if rng.random() < 0.5 else None
    shard_draw = (int(rng.integers(2, 6)), int(rng.choice([8, 16])), float(rng.random())) if case % 3 == 0 else None   # drawn here so that `only` replays a case exactly
    if only >= 0 and case != only:
        continue
    env = pkg.scenes.synthetic_hdri(256, 128, seed=env_seed) if env_seed is not None else None
    o = sc.upload(O.Oracle(sc.width, sc.height, sc.shadow_size, sc.max_lights))
    r = sc.upload(pkg.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
    r.set_option("keep_float_output", 1)
    if jitter:
        for name, value in hip_options.items(): r.set_option(name, value)
    if env is not None:
        o.create_hdri(env); r.create_hdri(env)
    o.pass_shadow_map(desc); o.pass_gbuffer(desc)
    o.set_precision(32); o.pass_shade(desc, settings, threads=threads); ldr32 = o.read_output()[0].copy()
    o.set_precision(64); o.pass_shade(desc, settings, threads=threads)
    img = r.render_frame(desc, settings)                      # visibility-plane path
    sm_ok = np.array_equal(o.read_shadow_map().view(np.uint32), r.read_shadow_map().view(np.uint32)) if sc.shadow_size else True
    og, hg = o.read_gbuffer(), r.read_gbuffer()
    gb_ok = all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(og, hg))
    oldr, _, orgba = o.read_output()
    hldr, _, hrgba = r.read_output()
    e_hip, e_f32 = np.abs(oldr - hldr).max(-1), np.abs(oldr - ldr32).max(-1)
    # conditioning of the formula itself: the float64 oracle on the same G-buffer with the tangent frame and the world position
    # (attributes 2..13) moved by ONE fp32 ulp per component, in four sign patterns.  At a grazing view (n.wo ~ 1e-5 under the normal
    # map) that alone moves the exact result by 1e-3: no fp32 evaluation -- whose normalised n carries half an ulp of rounding per
    # component -- can promise 1e-4 there, whether or not the oracle's own fp32 path happens to land close
    sens = np.zeros_like(e_hip)
    prng = np.random.default_rng(case)
    # one ulp OF THE VECTOR a component belongs to (tangent, bitangent, normal, position): the rounding error of a dot product scales
    # with its operands' magnitudes, and a component that is exactly 0 -- a floor at y = 0, an axis-aligned normal -- has no ulp of its own
    vec = og[0][..., 2:14].reshape(og[0].shape[:2] + (4, 3))
    step = np.repeat(np.spacing(np.abs(vec).max(-1, keepdims=True).astype(np.float32)), 3, axis=-1).reshape(og[0].shape[:2] + (12,))
    for pattern in range(4):
        up = np.ones(step.shape, bool) if pattern == 0 else (np.zeros(step.shape, bool) if pattern == 1 else prng.random(step.shape) < 0.5)
        pert = og[0].copy()
        pert[..., 2:14] = (pert[..., 2:14] + np.where(up, step, -step)).astype(np.float32)
        alt = o.shade_gbuffer(desc, settings, pert, og[1], threads=threads, want=("ldr",))["ldr"]
        sens = np.maximum(sens, np.abs(alt - oldr).max(-1))
    ill = (e_f32 > 5e-5) | (sens > 2.5e-5)                   # the literal fp32 evaluation misses the float64 value, or one input ulp moves it
    err = float(e_hip[~ill].max())
    ill_ok = bool((e_hip[ill] <= 4 * np.maximum(e_f32, sens)[ill]).all())
    n_ill, worst_ill = int(ill.sum()), float(e_hip[ill].max()) if ill.any() else 0.0
    mism = float((orgba != hrgba).mean())
    cov = float((og[1] != 0xFFFFFFFF).mean())
    # STRICT by default: EVERY pixel within 1e-4 of the float64 oracle, as in the pytest suite; the conditioning census above is reported, and only
    # ARCTIC_FUZZ_ALLOW_ILL=1 lets pixels it marks ill-conditioned pass at 4x the literal fp32 evaluation's own distance (scenes with grazing views under a normal map)
    strict = os.environ.get("ARCTIC_FUZZ_ALLOW_ILL") != "1"
    ok = sm_ok and gb_ok and err <= 1e-4 and (float(e_hip.max()) <= 1e-4 if strict else ill_ok) and np.abs(orgba.astype(int) - hrgba.astype(int)).max() <= 1 and np.array_equal(img, hrgba)
    shard_note = ""
    if ok and case % 3 == 0:   # a random interleaved shard of the same frame must reproduce its rows byte for byte
        from importlib import import_module
        sh = import_module("arctic_renderer_amd.sharding")
        world, band = shard_draw[0], shard_draw[1]
        k = min(world - 1, int(shard_draw[2] * world))
        rows = sh.owned_rows(sc.height, k, world, band)
        if len(rows):
            rs = sc.upload(pkg.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights, band_rows=band, shard=(k, world)))
            if env is not None:
                rs.create_hdri(env)
            if jitter:
                for name, value in hip_options.items(): rs.set_option(name, value)
            same = bool(np.array_equal(rs.render_frame(desc, settings), img[rows]))
            rs.close()
            ok = ok and same
            shard_note = f", shard {k}/{world} band {band} {'==' if same else '!='}"
    bad += not ok
    worst["ldr"], worst["rgba"] = max(worst["ldr"], err), max(worst["rgba"], mism)
    print(f"case {case:2d} config {cfg} {sc.width}x{sc.height} tm {settings[0]} sky {env is not None} coverage {cov:.2f}: shadow map {'==' if sm_ok else '!='}, "
          f"G-buffer {'==' if gb_ok else '!='}, max |ldr err| {err:.2e}, fp32-ill-conditioned pixels {n_ill} (HIP max {worst_ill:.1e}), rgba8 mismatch {mism:.1e}{shard_note} -> {'ok' if ok else 'FAIL'}", flush=True)
    if only >= 0 or not ok:   # a failing case dumps its worst pixels; `only` replays it alone
        ys, xs = np.nonzero(e_hip > 2e-5)
        print("pixels above 2e-5:", len(ys))
        _, ohdr, _ = o.read_output(); _, hhdr, _ = r.read_output()
        order = np.argsort(-e_hip[ys, xs])[:8]
        for y, x in zip(ys[order], xs[order]):
            a = og[0][y, x]
            print(f"  ({x},{y}) ldr o {oldr[y, x]} h {hldr[y, x]} hdr o {ohdr[y, x]} h {hhdr[y, x]} mat {og[1][y, x]} uv {a[0:2]} n {a[8:11]} world {a[11:14]} ls {a[14:18]}")
            # the same pixel through the oracle's literal fp32 arithmetic
            o.set_precision(32)
            one = o.shade_gbuffer(desc, settings, og[0][y:y + 1].copy(), og[1][y:y + 1].copy(), threads=1, want=("hdr", "ldr"))
            o.set_precision(64)
            print("     oracle fp32: ldr", one["ldr"][0, x], "hdr", one["hdr"][0, x])
    r.close(); o.close()
print(f"{n_cases} cases, {bad} failures; worst max |ldr err| {worst['ldr']:.2e}, worst rgba8 mismatch rate {worst['rgba']:.1e}")
sys.exit(1 if bad else 0)
