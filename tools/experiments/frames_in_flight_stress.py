"""frames in flight against one frame at a time, byte for byte: 24 frames per run with a random walk of the camera, objects moving now and
then and a shadow redraw in between, full-size configurations, three repetitions each (races, if any, show as differences)."""
import copy, sys, numpy as np, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import __graft_entry__ as e
pkg = e.load_package()
for cfg, scale in ((3, 1.0), (2, 1.0), (4, 0.5)):
    sc = pkg.scenes.CONFIGS[cfg](scale=scale)
    two = sc.upload(pkg.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
    one = sc.upload(pkg.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
    one.set_option("frames_in_flight", 1)
    n = 24
    outs = [[torch.empty((sc.height, sc.width, 4), dtype=torch.uint8, device="cuda") for _ in range(n)] for _ in range(2)]
    rng = np.random.default_rng(cfg)
    descs = []; desc = copy.deepcopy(sc.desc)
    for k in range(n):
        desc = copy.deepcopy(desc)
        e0 = np.array(desc.camera["eye"]); desc.camera["eye"] = tuple(float(v) for v in e0 + rng.uniform(-0.3, 0.3, 3))
        desc.camera["rotation"] = (float(desc.camera["rotation"][0] + rng.uniform(-3, 3)), float(desc.camera["rotation"][1] + rng.uniform(-8, 8)))
        if k % 5 == 3: desc.objects["trs"][int(rng.integers(0, len(desc.objects)))][12] += 0.25
        if k % 7 == 6: desc.sun["rotation"] = (float(desc.sun["rotation"][0] + 1.0), float(desc.sun["rotation"][1]))   # a shadow redraw in between
        descs.append(desc)
    for rep in range(4):
        if rep == 3:   # the last repetition redraws the shadow map in every frame: both shadow maps in flight all the time
            two.set_option("shadow_cache", 0); one.set_option("shadow_cache", 0)
        for k in range(n):
            for r, o in zip((two, one), outs): r.render_frame_device(descs[k], sc.settings, o[k].data_ptr())
        two.flush(); one.flush()
        same = all(bool((outs[0][k] == outs[1][k]).all().item()) for k in range(n))
        print(f"config {cfg} x{scale}: repetition {rep}: {n} frames in flight == one at a time: {same}", flush=True)
    two.close(); one.close()
