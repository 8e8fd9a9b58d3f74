#!/usr/bin/env python3
"""bench.py -- Mshaded-pixels/s of the forward PBR shading pass (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is ONE pass of the hot path (ps_main + post_process: G-buffer -> BRDF + PCF shadow +
point lights -> tonemap + gamma -> RGBA8) over one 4K frame's G-buffer, resident in HBM.  The
workload is BASELINE.json configs[2] (the config the metric is quoted on): the Sponza stand-in
at 3840x2160, 1 directional + 64 point lights, 4000^2 shadow map, ACES -- synthetic (the glTF
assets are not available offline), produced once, untimed, by the library's own shadow-map raster
and G-buffer prepass.

N > 1: the frame is sharded by rows, in interleaved bands of 16 rows dealt round-robin to the ranks
(lit regions are spatially clustered; contiguous ranges would be unbalanced); every step issues the
RCCL gather of the finished RGBA8 shards to rank 0 (the path's one real exchange step), which
overlaps the next step's shading through double buffering; total work is fixed -> "scaling": "strong".

The JSON line also carries
  roofline     achieved = 80 B x shaded pixels / mean time of the pass's two kernels (HIP events on
               the launch stream) against the 8 TB/s HBM peak; per-kernel times and rates next to it
               (k_light, 65 light evaluations per lit pixel, is FP32-VALU bound: SURVEY.md 7.3-1)
  cpu_baseline the CPU oracle (scalar C++ port of the same HLSL math) shading a bounded stripe of
               the SAME G-buffer on this host's cores -- a baseline, not the target.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

BYTES_PER_PIXEL = 80          # 72 B attributes + 4 B material id + 4 B RGBA8
HBM_PEAK_GBPS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
VALU_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: FP32 vector peak = every issue slot a v_pk_fma_f32 (4 flop per lane per 4-cycle slot)
# the packed light loop, from its ISA (make -C arctic-renderer_amd/csrc asm): per pair of lights 49 v_pk_* + ~4.5 plain VALU
# (4-cycle issue slots) + 6 transcendentals (v_rsq/v_rcp, 8 cycles = 2 slots each) = 65.5 slots -> 32.75 slots per light
# evaluation per lane, priced at the peak's 4 flop per slot
VALU_FLOPS_PER_LIGHT_EVAL = 32.75 * 4
FP32_PEAK_TFLOPS = 157.3      # vector FP32 (needs packed FMA)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, default=3, help="BASELINE.json config number (3 = the metric's config)")
    ap.add_argument("--scale", type=float, default=1.0, help="<1 shrinks the workload (debug only; makes the number invalid)")
    ap.add_argument("--cpu-rows", type=int, default=0, help="rows of the frame the CPU baseline shades (0 = auto)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--extras", action="store_true", help="also time the pass with culling off and count the executed light evaluations "
                    "(extra, slower launches of the same kernels: keep them out of a rocprofv3 --stats run of this command)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product has no CPU path")
    backend = os.environ.get("ARCTIC_BENCH_BACKEND", "nccl")    # "gloo" + ARCTIC_BENCH_SHARE_GPU=1: rehearse N ranks on one GPU
    if os.environ.get("ARCTIC_BENCH_SHARE_GPU") == "1":
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    pkg = entry.load_package()
    sharding = __import__("arctic_renderer_amd.sharding", fromlist=["x"])
    t0 = time.time()
    sc = pkg.scenes.CONFIGS[args.config](scale=args.scale)
    BAND = 16   # N > 1: rows are dealt to the ranks in interleaved bands of 16 (lit regions are clustered: load balance)
    if world > 1:
        r = sc.upload(pkg.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights, device=local, band_rows=BAND, shard=(rank, world)))
    else:
        r = sc.upload(pkg.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights, device=local))
    r.pass_shadow_map(sc.desc)      # untimed: the producers of the hot path's inputs
    r.pass_gbuffer(sc.desc)
    r.flush()
    if rank == 0:
        log(f"[bench] {sc.name}: {sc.width}x{sc.height}, {sc.n_triangles} triangles, {len(sc.materials)} materials, "
            f"{len(sc.lights)} point lights, shadow {sc.shadow_size}^2; setup {time.time() - t0:.1f}s")

    rows = r.rows
    # N > 1: the library runs on torch's stream, so the RCCL gather is stream-ordered after the shading pass with no
    # host synchronisation; two output buffers let the gather of frame k overlap the shading of frame k + 1
    # (each buffer is reused only after its own gather has completed).
    n_buf = 2 if world > 1 else 1
    # N > 1: every rank sends the same number of rows (its own, padded to the largest shard), so the exchange is ONE
    # ncclGather per frame; on the root the shards land back to back in a staging buffer and ONE indexed copy
    # de-interleaves them into the frame (the padding rows go to dummy rows past the frame's end: world * pad rows in all)
    pad, dest = sharding.padded_gather_plan(sc.height, world, BAND) if world > 1 else (rows, None)
    outs = [torch.empty((pad, sc.width, 4), dtype=torch.uint8, device="cuda") for _ in range(n_buf)]
    staging = [torch.empty((world * pad, sc.width, 4), dtype=torch.uint8, device="cuda") for _ in range(n_buf)] if (world > 1 and rank == 0) else None
    gathered = [[staging[b][k * pad:(k + 1) * pad] for k in range(world)] for b in range(n_buf)] if staging is not None else [None] * n_buf
    pending = [None] * n_buf
    frame_ext = torch.empty((world * pad, sc.width, 4), dtype=torch.uint8, device="cuda") if staging is not None else None
    frame = frame_ext[:sc.height] if frame_ext is not None else None
    perm = torch.as_tensor(dest, device="cuda") if staging is not None else None
    out_ptrs = [o.data_ptr() for o in outs]
    if world > 1:
        r.set_stream(torch.cuda.current_stream().cuda_stream)
    state = {"k": 0}
    shade = r.prepared_pass_shade(sc.desc, sc.settings)

    def step():
        b = state["k"] % n_buf
        state["k"] += 1
        if pending[b] is not None:
            finish(b)
        shade(out_ptrs[b])
        if world > 1:
            pending[b] = sharding.gather_rows(outs[b], gathered[b], rank, world, async_op=True)

    def finish(b):
        pending[b].wait()
        pending[b] = None
        if frame is not None:
            frame_ext.index_copy_(0, perm, staging[b])

    def drain():
        for b in range(n_buf):
            if pending[b] is not None:
                finish(b)

    # roofline: the shading pass is ONE kernel by default (k_material<2>: material fetch + shadow test + the light loop for the
    # lit pixels + tonemap + store; ARCTIC_OPT_LIGHT_PATH), each launch timed with HIP events on the library's stream
    # (measured first: it also brings clocks and caches to their steady state before the W warm-up steps)
    iters = max(10, min(args.steps, 50))
    ms = r.time_shade(sc.desc, sc.settings, warmup=20, iters=iters)

    for _ in range(args.warmup):
        step()
    drain()
    r.flush()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()
    r.flush()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t_start
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # outside the timed region: the exchange step alone (gathered bytes / gather time per link)
    gather_ms = None
    if world > 1:
        try:
            torch.cuda.synchronize(); dist.barrier()
            t0 = time.perf_counter()
            for _ in range(10):
                sharding.gather_rows(outs[0], gathered[0], rank, world)
            torch.cuda.synchronize(); dist.barrier()
            gather_ms = (time.perf_counter() - t0) / 10 * 1e3
        except Exception as exc:   # a measurement extra must never cost the bench line
            log(f"[bench] gather timing skipped: {exc}")
        r.set_stream(None)
    if world > 1 and rank == 0 and os.environ.get("ARCTIC_BENCH_VERIFY") == "1":
        # the multi-rank invariant, end to end: the gathered, de-interleaved frame == a single-device frame, byte for byte
        full = sc.upload(pkg.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights, device=local))
        ref = full.render_frame(sc.desc, sc.settings)
        full.close()
        same = bool((frame.cpu().numpy() == ref).all())
        log(f"[bench] verify: {world}-rank frame identical to the single-device frame: {same}")
        if not same:
            raise SystemExit("multi-rank frame differs from the single-device frame")
    # shaded pixels = pixels with geometry (100 % in this scene); counted, not assumed
    _, mat, _, _ = r.read_gbuffer(want=("material",))
    shaded_local = int((mat != 0xFFFFFFFF).sum())
    if world > 1:
        t = torch.tensor([shaded_local], dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t)
        shaded = int(t.item())
    else:
        shaded = shaded_local

    pass_ms = float(np.mean(ms))
    achieved = shaded_local * BYTES_PER_PIXEL / (pass_ms * 1e-3) / 1e9
    # lit pixels of this pass (read back from the stream counters; the kernels run exactly as in the timed loop)
    r.set_option("count_light_evals", 2)
    r.pass_shade(sc.desc, sc.settings)
    r.flush()
    lit_px = int(r.stats()[6])
    r.set_option("count_light_evals", 0)
    light_evals, ms_nocull, split = lit_px * len(sc.lights), None, None   # every lit pixel evaluates every point light (+ the sun)
    if args.extras:
        # not in the default run, so that a rocprofv3 --stats average over this command's launches is the timed kernel's:
        # the same pass with the exact culling disabled (every covered pixel evaluates the sun and all n_lights) ...
        r.set_option("culling", 0)
        ms_nocull = float(np.mean(r.time_shade(sc.desc, sc.settings, warmup=2, iters=10)))
        r.set_option("culling", 1)
        # ... the light evaluations actually executed, counted with atomics in the loop (wave-level skips included) ...
        r.set_option("count_light_evals", 1)
        r.pass_shade(sc.desc, sc.settings)
        r.flush()
        light_evals = int(r.stats()[5])
        r.set_option("count_light_evals", 0)
        # ... and the alternative path, k_material -> lit-pixel stream -> k_light, timed per kernel
        r.set_option("light_path", 1)
        a, b, c = r.time_shade_split(sc.desc, sc.settings, warmup=5, iters=20)
        split = [round(float(np.mean(x)), 4) for x in (a, b, c)]
        r.set_option("light_path", 0)
    result = None
    if rank == 0:
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc_path) and world == 1 and args.scale == 1.0 and args.config == 3:
            try:
                pmc = json.load(open(pmc_path))
                traffic = pmc.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        value = args.steps * shaded / dt / 1e6
        result = {
            "metric": "Mshaded-pixels/sec at 4K Sponza, 1 dir + 64 point lights; HBM GB/s vs roofline",
            "value": round(value, 1), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[{args.config - 1}]: {sc.name}, {sc.width}x{sc.height}, 1 dir + {len(sc.lights)} point lights, "
                                   f"shadow {sc.shadow_size}^2 PCF 5x5, tonemap {sc.settings[0]}, {len(sc.materials)} materials; shading pass over a resident G-buffer",
                       "triangles": sc.n_triangles, "shaded_pixels": shaded, "sharding": f"{BAND}-row bands round-robin over {world} ranks, RCCL gather to rank 0" if world > 1 else "none",
                       "scale": args.scale,
                       "exchange": None if gather_ms is None else {
                           "gather_ms": round(gather_ms, 4), "bytes_per_sender": int(pad) * sc.width * 4,
                           "GBps_per_link": round(int(pad) * sc.width * 4 / (gather_ms * 1e-3) / 1e9, 2),
                           "note": "one padded shard per rank into rank 0, timed alone (synchronous), outside the timed steps"}},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                         "kernel": "k_material<2> (material fetch + shadow test + packed light loop + tonemap: the whole pass)" if len(sc.lights) > 16 else "k_material<1> (the whole pass, scalar light loop)", "kernel_ms": round(pass_ms, 4),
                         "kernel_ms_p10_p50_p90": [round(float(np.percentile(ms, q)), 4) for q in (10, 50, 90)],
                         "bytes_per_pixel": BYTES_PER_PIXEL,
                         "lit_pixel_fraction": round(lit_px / max(shaded_local, 1), 4),
                         "point_light_evals_per_lit_pixel": round(light_evals / max(lit_px, 1), 2),
                         "point_light_evals_per_pixel": round(light_evals / max(shaded_local, 1), 2),
                         "Gevals_per_s": round(light_evals / (pass_ms * 1e-3) / 1e9, 1),
                         # the other roof: the FP32 vector peak.  The kernel has a memory-bound part (every pixel) and
                         # a VALU-bound part (the light loop of the lit quarter); this is the loop's share of ALL issue slots of the launch
                         "valu": {"achieved": round(light_evals * VALU_FLOPS_PER_LIGHT_EVAL / (pass_ms * 1e-3) / 1e12, 1),
                                  "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s (FMA = 2, issue-slot equivalents of the light loop)",
                                  "frac": round(light_evals * VALU_FLOPS_PER_LIGHT_EVAL / (pass_ms * 1e-3) / 1e12 / VALU_PEAK_TFLOPS, 4)},
                         "kernel_ms_no_culling": round(ms_nocull, 4) if ms_nocull else None,
                         "achieved_no_culling": round(shaded_local * BYTES_PER_PIXEL / (ms_nocull * 1e-3) / 1e9, 1) if ms_nocull else None,
                         "stream_path_ms_pass_material_light": split},
        }
        if world == 1 and not args.no_cpu:
            result["cpu_baseline"] = cpu_baseline(pkg, sc, r, args.cpu_rows)
    r.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)


def cpu_baseline(pkg, sc, r, cpu_rows):
    """the CPU oracle (test infrastructure; here only as the timed baseline) shading a bounded stripe
    of the same G-buffer and shadow map, on all host threads."""
    from oracle import oracle as O
    O.build()
    threads = O.hardware_threads() or os.cpu_count() or 1
    attrs, mat, _, _ = r.read_gbuffer(want=("attrs", "material"))
    o = sc.upload(O.Oracle(sc.width, sc.height, sc.shadow_size, sc.max_lights))
    if sc.shadow_size:
        o.write_shadow_map(r.read_shadow_map())
    o.set_precision(32)     # the literal fp32 restatement of the HLSL: what a CPU port of the shader would run
    # calibrate on 8 rows, then size the sample for ~15 s of wall time (whole frame, repeated, if the host is fast)
    rows0 = min(8, attrs.shape[0])
    mid = attrs.shape[0] // 2
    t = time.perf_counter()
    o.shade_gbuffer(sc.desc, sc.settings, attrs[mid:mid + rows0], mat[mid:mid + rows0], threads=threads, want=("rgba8",))
    per_row = (time.perf_counter() - t) / rows0
    n = cpu_rows or int(max(8, min(attrs.shape[0], 15.0 / max(per_row, 1e-6))))
    n = max(4, n // 4 * 4)
    start = max(0, mid - n // 2)
    reps, dt = 0, 0.0
    t = time.perf_counter()
    while reps < 1 or (dt < 10.0 and reps < 50):
        o.shade_gbuffer(sc.desc, sc.settings, attrs[start:start + n], mat[start:start + n], threads=threads, want=("rgba8",))
        reps += 1
        dt = time.perf_counter() - t
    px = int((mat[start:start + n] != 0xFFFFFFFF).sum())
    # one thread, on a few rows (SURVEY 8d asks for both figures)
    n1, reps1, dt1 = min(4, attrs.shape[0]), 0, 0.0
    t = time.perf_counter()
    while reps1 < 1 or (dt1 < 2.0 and reps1 < 100):
        o.shade_gbuffer(sc.desc, sc.settings, attrs[mid:mid + n1], mat[mid:mid + n1], threads=1, want=("rgba8",))
        reps1 += 1
        dt1 = time.perf_counter() - t
    one_thread = int((mat[mid:mid + n1] != 0xFFFFFFFF).sum()) * reps1 / dt1 / 1e6
    o.close()
    model = ""
    try:
        model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except Exception:
        pass
    return {"value": round(px * reps / dt / 1e6, 3), "unit": "Mpixels/s", "cores": threads, "kind": "port",
            "one_thread_value": round(one_thread, 4), "cpu_model": model,
            "sample": f"rows {start}..{start + n} of the {sc.height}-row frame x {reps} ({px * reps} shaded pixels, {dt:.1f} s), same "
                      f"G-buffer, shadow map and {len(sc.lights)} lights; scalar fp32 C++ oracle, {threads} threads, no culling"}


if __name__ == "__main__":
    main()
