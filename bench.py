#!/usr/bin/env python3
"""bench.py -- Mshaded-pixels/s of the forward PBR shading pass (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Both forms run N ranks, one per GPU: under a launcher (WORLD_SIZE in the environment) this process is one of them; without one
and N > 1 it starts the N ranks itself (launch_ranks: child processes, before any GPU call), relays rank 0's JSON line and exits
non-zero if any rank fails.  The first multi-rank step is compared byte for byte with a single-device frame before anything is timed.

A "step" is ONE pass of the hot path (ps_main + post_process: G-buffer -> BRDF + PCF shadow +
point lights -> tonemap + gamma -> RGBA8) over one 4K frame's G-buffer, resident in HBM.  The
workload is BASELINE.json configs[2] (the config the metric is quoted on): the Sponza stand-in
at 3840x2160, 1 directional + 64 point lights, 4000^2 shadow map, ACES -- synthetic (the glTF
assets are not available offline), produced once, untimed, by the library's own shadow-map raster
and G-buffer prepass.

N > 1: the frame is sharded by rows, in interleaved bands of 16 rows dealt round-robin to the ranks
(lit regions are spatially clustered; contiguous ranges would be unbalanced); every step issues the
gather of the finished RGBA8 shards to rank 0 (the path's one real exchange step) through the C-ABI
(arctic_gather_frame, include/arctic_dist.h: RCCL send/recv on the handle's communication stream + one
placement kernel on the root), which overlaps the next step's shading through two shard buffers; total
work is fixed -> "scaling": "strong".  torch.distributed only carries the 128-byte communicator id and
the timing barriers (ARCTIC_BENCH_EXCHANGE=torch keeps the round-1 path: dist.gather + index_copy_).

The JSON line also carries
  roofline     the dominant kernel = the pass itself (ONE launch: k_material<loop>): achieved = 80 B x shaded pixels /
               kernel_ms, kernel_ms = HIP events on the launch stream around the K timed steps / K (the same launches
               ms_per_step is the wall clock of).  Which roof binds is DERIVED, not assumed: valu_issue_frac = the kernel's
               vector-issue cycles / (1024 SIMDs x 2.4 GHz x kernel_ms), priced per instruction class (the light loop's v_pk_*
               from the executed pair trips x the census of its ISA, transcendentals from their counter, the rest by the static
               mix of the kernel's other code; costs from tools/experiments/valu_rates.hip); valu_flop_frac = FP32 flops
               (FMA = 2) / kernel_ms / 157.3 TFLOP/s; hbm_frac = real HBM bytes / kernel_ms / 8 TB/s.  The counter inputs are
               STATIC, read from profiles/pmc_latest.json (rocprofv3 --pmc in separate passes, tools/profile_round.sh) and
               profiles/isa_census_latest.json (make census), and labelled so.  Also in the line: the same pass with exact
               culling off (every pixel lit) and with 16 / 4 / 0 point lights (run after the timed loop; --no-extras
               leaves them out, which is what a rocprofv3 --stats run of this command wants).
  settle       before anything is measured the pass is repeated, untimed, for ARCTIC_BENCH_SETTLE_MS (250): a device that has
               been idle runs its first ~70 ms of load at other clocks than it sustains (0.22 against 0.19 ms per pass).
  cpu_baseline the CPU oracle (scalar C++ port of the same HLSL math) shading a bounded stripe of
               the SAME G-buffer on this host's cores -- a baseline, not the target.
"""
import argparse
import json
import os
import sys
import time

# The HSA runtime waits for completion signals by polling instead of sleeping on an interrupt (set before anything initialises
# it; a caller's own setting wins): measured on this pool with the driver's command, 0.1968 -> 0.1933 ms per step and, between
# the HIP events around the same launches, 0.1948 -> 0.1910 -- back-to-back launches are handed over faster, and the timed
# region of 20 steps ends ~40 us earlier.  A host-side choice (a core per rank spins while it waits); recorded in the line.
os.environ.setdefault("HSA_ENABLE_INTERRUPT", "0")
# ... and kernel arguments live in device memory (the default of this ROCm; with 0 -- arguments fetched from host memory --
# whole frames measured 0.281 -> 0.292 ms and the pass 0.195 -> 0.198: the shading kernels re-read their 400 bytes per tile)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

BYTES_PER_PIXEL = 80          # 72 B attributes + 4 B material id + 4 B RGBA8
HBM_PEAK_GBPS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
N_SIMDS = 1024                # 256 CUs x 4
CLOCK_NOMINAL_GHZ = 2.4       # MI355X_MICROARCH.md: max clock
FP32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: vector FP32 (256 CUs x 4 SIMDs x 32 lanes x 2 flops x 2.4 GHz)
# vector-issue cost in cycles per wave64 instruction, by class (profiles/r2_valu_rates.txt, W >= 4 waves per SIMD, at the 2.4 GHz
# the table is normalised to): v_pk_* 4.4; plain v_add / v_mul / v_fma / v_mov / v_and / shifts 2.6 ("fast"); min / max / cvt / cmp /
# cndmask / bfe / sdwa 4.2 ("slow"); transcendentals 8.  Which instruction is of which class: the census of the kernel's ISA
# (make -C arctic-renderer_amd/csrc census -> profiles/isa_census_latest.json): the light loop's body exactly, the rest by its static mix.
CYCLES = {"pk": 4.4, "fast": 2.6, "slow": 4.2, "trans": 8.0}
SETTLE_MS = float(os.environ.get("ARCTIC_BENCH_SETTLE_MS", "250"))   # untimed: the pass back to back until the device has left its idle clocks


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def count_gpus_without_hip():
    """GPUs this process may use, WITHOUT a HIP call (the launcher must not initialise the GPU): the KFD topology's nodes that have SIMDs
    and whose render node this process may open (a lease that restricts devices by cgroup / device-node permissions rather than by
    environment still lists every GPU in the topology), cut down by the *_VISIBLE_DEVICES lists that are set -- they compose (HIP indexes
    into what ROCr shows), so the smallest count holds.  0 without an amdgpu compute driver; None when the topology cannot be read: the
    ranks then fail by themselves, loudly, if a device is missing."""
    listed = []
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            listed.append(len([x for x in v.split(",") if x.strip() != ""]))
    base = "/sys/class/kfd/kfd/topology/nodes"
    n = None
    if not os.path.isdir("/sys/class/kfd"):
        n = 0        # no amdgpu compute driver on this machine at all
    else:
        try:
            n = 0
            for node in os.listdir(base):
                with open(os.path.join(base, node, "properties")) as f:
                    props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
                if int(props.get("simd_count", "0")) <= 0:
                    continue
                minor = int(props.get("drm_render_minor", "-1"))
                dev = f"/dev/dri/renderD{minor}"
                if minor >= 0 and os.path.exists("/dev/dri") and not (os.path.exists(dev) and os.access(dev, os.R_OK | os.W_OK)):
                    continue     # in the topology, but not this process's to open
                n += 1
        except (OSError, ValueError):
            n = None
    if listed:
        return min(listed) if n is None else min([n] + listed)
    return n


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher (no WORLD_SIZE in the environment): start the N ranks here -- one child
    process per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set the way torch.distributed.run sets them --
    BEFORE this process has made any GPU call (it never makes one, not even to count the devices -- count_gpus_without_hip: a
    process that has initialised the GPU must not fork + exec on this pool; under a profiler, whose preloaded library has
    initialised it already, the launcher refuses).  Rank 0's stdout (the one JSON line) is relayed; the exit code is non-zero when any rank fails, and the
    other ranks are then stopped by their exact pids (they may be blocked in a collective)."""
    import socket
    import subprocess
    import threading
    share = os.environ.get("ARCTIC_BENCH_SHARE_GPU") == "1"
    if any("rocprof" in os.environ.get(v, "") for v in ("LD_PRELOAD", "HSA_TOOLS_LIB", "ROCP_TOOL_LIBRARIES")) or os.environ.get("ROCPROFILER_REGISTER_FORCE_LOAD"):
        # under rocprofv3 the profiler's preloaded library has initialised the GPU before this program started: starting the ranks
        # from here would be a fork + exec from a GPU-initialised process (refused on this pool)
        raise SystemExit("bench.py --gpus N under a profiler: profile ONE rank directly (RANK / WORLD_SIZE / MASTER_* set by hand), not the launcher")
    if not share:
        have = count_gpus_without_hip()
        if have is not None and have < n:
            raise SystemExit(f"bench.py --gpus {n}: this node shows {have} HIP device(s); the multi-GPU leg needs one GPU per rank "
                             f"(a rehearsal of the N-rank code path on one GPU: ARCTIC_BENCH_BACKEND=gloo ARCTIC_BENCH_SHARE_GPU=1)")
    port = os.environ.get("MASTER_PORT")
    if not port:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = str(s.getsockname()[1])
        s.close()
    procs, lines = [], []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port, ARCTIC_BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on this pool
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if rank == 0 else sys.stderr, text=(rank == 0)))
    reader = threading.Thread(target=lambda: lines.extend(procs[0].stdout), daemon=True)
    reader.start()
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        time.sleep(0.1)
        failed = next((k for k, p in enumerate(procs) if p.poll() not in (None, 0)), None)
    if failed is None:
        failed = next((k for k, p in enumerate(procs) if p.returncode != 0), None)
    if failed is not None:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_end = time.time() + 10
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
        log(f"[bench] rank {failed} exited with code {procs[failed].returncode}; the other ranks were stopped")
        raise SystemExit(procs[failed].returncode or 1)
    reader.join(timeout=10)
    out = [l for l in lines if l.strip()]
    if not out:
        raise SystemExit("[bench] rank 0 printed no result line")
    sys.stdout.write(out[-1] if out[-1].endswith("\n") else out[-1] + "\n")
    sys.stdout.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, default=3, help="BASELINE.json config number (3 = the metric's config)")
    ap.add_argument("--scale", type=float, default=1.0, help="<1 shrinks the workload (debug only; makes the number invalid)")
    ap.add_argument("--cpu-rows", type=int, default=0, help="rows of the frame the CPU baseline shades (0 = auto)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed extra passes (culling off, 16 / 4 / 0 lights, light statistics): "
                    "for a rocprofv3 --stats run of this command, whose kernel average should be the timed launches'")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus, sys.argv[1:])      # no launcher around us: be the launcher (no GPU call in this process)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"warning: --gpus {args.gpus} but the launcher set WORLD_SIZE={world}; running {world} ranks (n_gpus in the line is {world})")
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
    tiling = os.environ.get("ARCTIC_BENCH_TEXTURE_TILING")   # A/B only: ARCTIC_OPT_TEXTURE_TILING for the materials (default: the library's choice by image size)

    def new_handle(**kw):
        h = pkg.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights, device=local, **kw)
        if tiling is not None:
            h.set_option("texture_tiling", int(tiling))
        return sc.upload(h)
    r = new_handle(band_rows=BAND, shard=(rank, world)) if world > 1 else new_handle()
    # the library launches on torch's stream (torch events see its kernels) -- set before the communicator is made: the exchange then
    # runs on the handle's own stream, idle from here on, instead of a stream more (hardware queues are few: DESIGN.md 4.3)
    r.set_stream(torch.cuda.current_stream().cuda_stream)
    tile_order = 1 if os.environ.get("ARCTIC_BENCH_TILE_ORDER") == "1" else 0   # ARCTIC_OPT_TILE_ORDER: off like the library's default (A/B: =1)
    r.set_option("tile_order", tile_order)
    cabi = False
    if world > 1 and backend == "nccl" and os.environ.get("ARCTIC_BENCH_EXCHANGE", "cabi") == "cabi":
        # The exchange below Python: an RCCL communicator owned by the handle; the 128-byte id travels over torch.distributed.
        # Every step here is a collective that EVERY rank executes whatever happened before it on that rank (a rank that skipped
        # one would leave the others blocked in it): broadcast (None when rank 0 could not make an id) -> init -> vote.
        uid = None
        if rank == 0:
            try:
                uid = pkg.Renderer.comm_unique_id()
            except Exception as exc:
                log(f"[bench] rank 0: no RCCL unique id ({exc})")
        box = [uid]
        dist.broadcast_object_list(box, src=0)
        mine = 0
        if box[0] is not None:
            try:
                r.comm_init(box[0], rank, world)
                mine = 1
            except Exception as exc:
                log(f"[bench] rank {rank}: arctic_comm_init failed ({exc})")
        flag = torch.tensor([mine], device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        cabi = bool(flag.item())
        if cabi:
            r.set_option("shadow_sharded", 1)
        else:       # lost vote: no rank keeps a communicator (a sharded shadow pass would all-gather with ranks that have none)
            log(f"[bench] rank {rank}: C-ABI exchange not available on every rank; falling back to torch.distributed")
            try:
                r.comm_destroy()
            except Exception:
                pass
            r.set_option("shadow_sharded", 0)
    r.pass_shadow_map(sc.desc)      # untimed: the producers of the hot path's inputs
    r.pass_gbuffer(sc.desc)
    r.flush()
    # a working set below 512 MiB (configs 1 and 2: 1080p is 166 MB of G-buffer + output) would be served by the 256 MiB
    # Infinity Cache if the same buffers were shaded again and again -- the timed steps then ROTATE over three handles, each with its own
    # G-buffer, shadow map, textures and output (a real frame's G-buffer has just been written and is as large as the cache at most once)
    working_set = sc.width * sc.height * (BYTES_PER_PIXEL - 4 + 4)
    rotation = [r]
    if world == 1 and working_set < (512 << 20) and os.environ.get("ARCTIC_BENCH_ROTATE", "1") != "0":
        for _ in range(2):
            h = new_handle()
            h.set_stream(torch.cuda.current_stream().cuda_stream)
            h.set_option("tile_order", tile_order)
            h.pass_shadow_map(sc.desc)
            h.pass_gbuffer(sc.desc)
            h.flush()
            rotation.append(h)
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
    pad, dest = sharding.padded_gather_plan(sc.height, world, BAND) if (world > 1 and not cabi) else (rows, None)
    n_buf = max(n_buf, len(rotation))
    outs = [torch.empty((pad, sc.width, 4), dtype=torch.uint8, device="cuda") for _ in range(n_buf)]
    staging = [torch.empty((world * pad, sc.width, 4), dtype=torch.uint8, device="cuda") for _ in range(n_buf)] if (world > 1 and rank == 0 and not cabi) else None
    gathered = [[staging[b][k * pad:(k + 1) * pad] for k in range(world)] for b in range(n_buf)] if staging is not None else [None] * n_buf
    pending = [None] * n_buf
    frame_ext = torch.empty((world * pad, sc.width, 4), dtype=torch.uint8, device="cuda") if staging is not None else None
    frame = frame_ext[:sc.height] if frame_ext is not None else None
    if cabi and rank == 0:
        frame = torch.empty((sc.height, sc.width, 4), dtype=torch.uint8, device="cuda")
    frame_ptr = frame.data_ptr() if (cabi and rank == 0) else None
    perm = torch.as_tensor(dest, device="cuda") if staging is not None else None
    out_ptrs = [o.data_ptr() for o in outs]
    r.set_stream(torch.cuda.current_stream().cuda_stream)   # the library launches on torch's stream: torch events see its kernels
    state = {"k": 0, "launches": 0, "t_first": None}   # launches: passes enqueued by this process so far (a profiled run's kernel trace is cut with it)
    shade_raw = [h.prepared_pass_shade(sc.desc, sc.settings) for h in rotation]

    def shade(ptr):
        if state["t_first"] is None:
            state["t_first"] = time.perf_counter()
        shade_raw[state["launches"] % len(rotation)](ptr)      # (one handle, or three in turn: see `rotation`)
        state["launches"] += 1

    def step():
        b = state["k"] % n_buf
        state["k"] += 1
        if pending[b] is not None:
            finish(b)
        shade(out_ptrs[b])      # (waits by itself for the gather that last read this buffer)
        if cabi:
            r.gather_frame(out_ptrs[b], frame_ptr, 0)
        elif world > 1:
            pending[b] = sharding.gather_rows(outs[b], gathered[b], rank, world, async_op=True, equal_rows=True)

    def finish(b):
        pending[b].wait()
        pending[b] = None
        if frame is not None:
            frame_ext.index_copy_(0, perm, staging[b])

    def drain():
        for b in range(n_buf):
            if pending[b] is not None:
                finish(b)

    verified = None
    if world > 1 and os.environ.get("ARCTIC_BENCH_VERIFY", "1") != "0":
        # The multi-rank invariant, end to end, on the FIRST multi-rank step and before anything is timed: the gathered,
        # de-interleaved frame == a single-device frame, byte for byte.  A mismatch ends the run on every rank (non-zero exit):
        # an unverified exchange path is never timed.  (ARCTIC_BENCH_VERIFY=0 skips it.)
        step(); drain(); r.flush(); torch.cuda.synchronize()
        same = 1
        if rank == 0:
            full = sc.upload(pkg.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights, device=local))
            ref = full.render_frame(sc.desc, sc.settings)
            full.close()
            same = int(bool((frame.cpu().numpy() == ref).all()))
            log(f"[bench] verify: {world}-rank frame identical to the single-device frame: {bool(same)}")
        t = torch.tensor([same], dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if not int(t.item()):
            r.close()
            raise SystemExit("multi-rank frame differs from the single-device frame")
        verified = True

    def timed_region(k_steps, w_steps):
        """W untimed steps, then exactly K steps between barrier + synchronize on both sides; returns (wall seconds, HIP-event ms, index of
        the first timed launch)"""
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(w_steps):
            step()
        drain()
        r.flush()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        first = state["launches"]
        t_start = time.perf_counter()
        e0.record()
        for _ in range(k_steps):
            step()
        e1.record()
        drain()
        torch.cuda.synchronize()   # (every stream of the device: the library's too)
        if world > 1:
            dist.barrier()
        wall = time.perf_counter() - t_start
        r.flush()                  # the library's own synchronising call reports what a pass may have flagged (outside the timed region: it adds nothing to wait for)
        return wall, e0.elapsed_time(e1), first, t_start

    # COLD figure first (rounds 1-2 measured this way; kept so that rounds stay comparable from the driver's record alone): the same
    # W + K steps straight after set-up, before the device has been kept busy for any length of time.
    cold = None
    if os.environ.get("ARCTIC_BENCH_COLD", "1") != "0":
        c_wall, c_ev, _, _ = timed_region(args.steps, args.warmup)
        if world > 1:
            t = torch.tensor([c_wall], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            c_wall = float(t.item())
        cold = {"ms_per_step": round(c_wall / args.steps * 1e3, 4), "kernel_ms": round(c_ev / args.steps, 4),
                "note": f"the same {args.warmup} + {args.steps} steps timed the same way straight after set-up, BEFORE the settle loop (how rounds 1-2 were measured)"}

    # A device that has been idle runs its first ~100 ms of work at other clocks than the ones it sustains (measured on this pool:
    # the same pass 0.22 ms in the first 20 ms of load, 0.19 ms after 70 ms, then stable to 0.3 %): the pass is repeated, untimed,
    # for SETTLE_MS before anything is measured.  The timed region below is still exactly W warm-up steps + K steps.
    if SETTLE_MS > 0:
        t_settle = time.perf_counter()
        while (time.perf_counter() - t_settle) * 1e3 < SETTLE_MS:
            for _ in range(50):
                shade(out_ptrs[0])
            r.flush()
    # isolated launches, each between its own pair of HIP events (p10/p50/p90 in the line)
    iters = max(10, min(args.steps, 50))
    ms = r.time_shade(sc.desc, sc.settings, warmup=20, iters=iters)
    state["launches"] += 20 + iters

    dt, ev_ms, first_timed, t_start = timed_region(args.steps, args.warmup)
    warmup_effective = {"launches": first_timed, "ms": round((t_start - (state["t_first"] or t_start)) * 1e3, 1),
                        "note": "passes this process had enqueued, and wall time since the first of them, when the timed region began: cold region + settle loop + isolated launches + --warmup"}
    kernel_ms = ev_ms / args.steps   # HIP events on the launch stream around the K timed launches
    kernel_ms_source = "HIP events on the launch stream around the K timed steps / K"
    if world > 1:   # there the span between the events includes the waits for the gathers that free the shard buffers: not kernel time
        kernel_ms = float(np.median(ms))
        kernel_ms_source = "median of this rank's isolated launches (its own shard), each between its own pair of HIP events"
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
                if cabi:
                    r.gather_frame(out_ptrs[0], frame_ptr, 0)
                else:
                    sharding.gather_rows(outs[0], gathered[0], rank, world, equal_rows=True)
            r.flush(); torch.cuda.synchronize(); dist.barrier()
            gather_ms = (time.perf_counter() - t0) / 10 * 1e3
        except Exception as exc:   # a measurement extra must never cost the bench line
            log(f"[bench] gather timing skipped: {exc}")
    # whole frames, outside the timed region (shadow raster [only when the sun moves] + visibility prepass of this rank's rows +
    # shading from the visibility plane [+ gather]): what an application sees, next to the pass the metric is defined on
    whole = {}
    try:
        for name, cache in (("static_sun", 1), ("moving_sun", 0)):
            r.set_option("shadow_cache", cache)
            for k in range(3 + 20):
                if k == 3:
                    r.flush(); torch.cuda.synchronize()
                    if world > 1:
                        dist.barrier()
                    t0 = time.perf_counter()
                r.render_frame_device(sc.desc, sc.settings, out_ptrs[k % n_buf])
                if cabi:
                    r.gather_frame(out_ptrs[k % n_buf], frame_ptr, 0)
            r.flush(); torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            whole[name + "_ms"] = round((time.perf_counter() - t0) / 20 * 1e3, 4)
        r.set_option("shadow_cache", 1)
        whole["note"] = ("arctic_render_frame_device per rank, 20 frames enqueued back to back (static sun: two frames in flight; moving sun: shadow pass beside the visibility prepass)" + (" + arctic_gather_frame; the shadow map is drawn in light-space row shards "
                         "and all-gathered when the sun moves" if cabi else "") + "; vertex transform and triangle setup of the whole scene are "
                         "redundant per rank (DESIGN.md section 5)")
    except Exception as exc:
        log(f"[bench] whole-frame timing skipped: {exc}")
    r.set_stream(None)
    # shaded pixels = pixels with geometry (100 % in this scene); counted, not assumed
    _, mat, _, _ = r.read_gbuffer(want=("material",))
    shaded_local = int((mat != 0xFFFFFFFF).sum())
    if world > 1:
        t = torch.tensor([shaded_local], dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t)
        shaded = int(t.item())
    else:
        shaded = shaded_local

    achieved = shaded_local * BYTES_PER_PIXEL / (kernel_ms * 1e-3) / 1e9
    n_lights = len(sc.lights)
    extras = {}
    lit_px = None
    if not args.no_extras and world == 1:
        # untimed, after the timed loop.  Light statistics from the counting variant of the kernel:
        r.set_option("count_light_evals", 1)
        r.pass_shade(sc.desc, sc.settings)
        r.flush()
        st = [int(x) for x in r.stats()]
        r.set_option("count_light_evals", 0)
        lit_px = st[6]
        extras["light_stats"] = {"lit_pixels": st[6], "point_light_evals": st[5], "evals_with_n_dot_wi_gt_0": st[7],
                                 "zero_contribution_fraction": round(1.0 - st[7] / max(st[5], 1), 4),
                                 "lit_tiles": st[9], "tile_light_pairs_all_zero": st[8],
                                 "tile_cullable_fraction": round(st[8] / max(st[9] * n_lights, 1), 4),
                                 "note": "exact culling = n.wi <= 0 (forward.hlsl:191-192); tile_cullable = what a per-tile light list could skip"}
        # the same pass with every pixel lit (exact culling of fully shadowed pixels off) ...
        r.set_option("culling", 0)
        ms_all = float(np.mean(r.time_shade(sc.desc, sc.settings, warmup=3, iters=10)))
        r.set_option("culling", 1)
        extras["all_pixels_lit"] = {"kernel_ms": round(ms_all, 4), "frac": round(shaded_local * BYTES_PER_PIXEL / (ms_all * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
        # ... and the same G-buffer with the reference's own light counts (MAX_NUM_POINT_LIGHTS = 16) and fewer: the memory-bound regime
        # (16 lights: the packed loop, 4 and 0: the scalar one -- the automatic choice switches at 12)
        for n in (16, 4, 0):
            if n < n_lights:
                r.update_lights(sc.lights[:n])
                m = float(np.mean(r.time_shade(sc.desc, sc.settings, warmup=3, iters=20)))
                extras[f"{n}_point_lights"] = {"kernel_ms": round(m, 4), "frac": round(shaded_local * BYTES_PER_PIXEL / (m * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
        r.update_lights(sc.lights)
    result = None
    if rank == 0:
        pmc, pmc_src = None, None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc_path) and world == 1 and args.scale == 1.0 and args.config == 3:
            try:
                pmc, pmc_src = json.load(open(pmc_path)), "profiles/pmc_latest.json"
            except Exception:
                pmc = None
        traffic = pmc.get("hbm_bytes_per_launch") if pmc else None
        build_now, build_pmc = entry.source_id(), (pmc or {}).get("build") or {}
        if pmc and build_pmc.get("csrc_sha16") != build_now["csrc_sha16"]:
            log(f"[bench] warning: profiles/pmc_latest.json was measured on other kernel sources (csrc {build_pmc.get('csrc_sha16')}, commit {build_pmc.get('commit')}) "
                f"than this run's (csrc {build_now['csrc_sha16']}): roofline.traffic / valu_issue_frac are STATIC figures of that build")
        roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                "traffic_source": f"static: {pmc_src}, round {pmc.get('round')} (rocprofv3 --pmc, separate passes; not measured in this run)" if pmc else None,
                "traffic_commit": build_pmc.get("commit") if pmc else None,
                "traffic_sources_match": (build_pmc.get("csrc_sha16") == build_now["csrc_sha16"]) if pmc else None,
                "build": build_now,
                "tile_order": tile_order,
                "kernel": f"k_material<{2 if n_lights > 12 else 1}> (the whole pass in one launch: material fetch, shadow test, "
                          f"{'packed' if n_lights > 12 else 'scalar'} light loop, tonemap, store; {'two tiles' if sc.width * sc.height >= 3000000 else 'one tile'} per wave, "
                          + ("strips handed out in the order the G-buffer pass left (ARCTIC_OPT_TILE_ORDER=1: its one-workgroup order kernel costs the UNTIMED G-buffer pass ~112 us at 4K))"
                             if tile_order else "tile rows dealt to the XCDs by row, ARCTIC_OPT_TILE_ORDER=0: the library's default)"),
                "settle_ms_before_measuring": SETTLE_MS,
                "warmup_effective": warmup_effective,
                "timed_launches": [first_timed, first_timed + args.steps],
                "cold": None if cold is None else dict(cold, frac=round(shaded_local * BYTES_PER_PIXEL / (cold["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)),
                "kernel_ms": round(kernel_ms, 4),
                "kernel_ms_source": kernel_ms_source,
                "isolated_launch_ms_p10_p50_p90": [round(float(np.percentile(ms, q)), 4) for q in (10, 50, 90)],
                "bytes_per_pixel": BYTES_PER_PIXEL,
                "rotation": {"sets": len(rotation), "working_set_bytes": working_set,
                             "note": "timed steps rotate over this many handles (own G-buffer, shadow map, textures, output each) when the working set is below "
                                     "512 MiB, so that the 256 MiB Infinity Cache cannot serve re-reads (SURVEY 8d); the isolated launches and the extras use ONE handle"}}
        if lit_px is not None:
            roof["lit_pixel_fraction"] = round(lit_px / max(shaded_local, 1), 4)
            roof["Gevals_per_s"] = round(lit_px * n_lights / (kernel_ms * 1e-3) / 1e9, 1)
        census = None
        try:
            census = json.load(open(os.path.join(ROOT, "profiles", "isa_census_latest.json")))
        except Exception:
            census = None
        if pmc and pmc.get("SQ_INSTS_VALU"):
            # which roof binds, from the counters: vector-issue cycles against the cycles the launch had, real HBM bytes against 8 TB/s.
            # Instruction classes: the light loop's v_pk_* from the executed pair trips (light_stats: lit tiles x pairs) x the census
            # of its body; transcendentals from their own counter; the remainder of SQ_INSTS_VALU priced with the static mix of the
            # kernel's other code.
            total, trans = pmc["SQ_INSTS_VALU"], pmc.get("trans_insts", 0.0)
            ls = extras.get("light_stats")
            if census and ls and n_lights > 12:
                body = census["per_pair_trip"]
                trips = ls["lit_tiles"] * ((n_lights + 1) // 2)
                pk_loop = trips * (body["pk_fma"] + body["pk_other"])
                rest = max(total - trans - pk_loop, 0.0)
                mix = census["rest_mix"]
                cyc = pk_loop * CYCLES["pk"] + trans * CYCLES["trans"] + rest * (mix["fast"] * CYCLES["fast"] + mix["slow"] * CYCLES["slow"] + mix["pk"] * CYCLES["pk"])
                # flops: FMA = 2; per lane and pair trip 4 per v_pk_fma, 2 per other v_pk, 1 per transcendental
                flops = 64.0 * (trips * (4 * body["pk_fma"] + 2 * body["pk_other"] + body["trans"]) + rest * mix["flops_per_inst_lane"])
                pricing = {"light_loop_pk_insts": pk_loop, "other_valu_insts": rest, "cycles_per_inst": CYCLES,
                           "other_mix_fast_slow_pk": [round(mix["fast"], 3), round(mix["slow"], 3), round(mix["pk"], 3)],
                           "source": f"static: {pmc_src} (counters), profiles/isa_census_latest.json (classes), light_stats of this run (trips)"}
                roof["valu_flop_frac"] = {"value": round(flops / (kernel_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4), "TFLOPs": round(flops / (kernel_ms * 1e-3) / 1e12, 1),
                                          "peak_TFLOPs": FP32_PEAK_TFLOPS, "flops_per_launch": flops,
                                          "note": "FMA = 2 flops; light loop from its ISA census x executed pair trips, the rest from SQ_INSTS_VALU x the static flop mix of the other code"}
            else:   # no census / no light statistics (--no-extras) / scalar loop: every non-transcendental instruction at the slow-class cost
                cyc = (total - trans) * CYCLES["slow"] + trans * CYCLES["trans"]
                pricing = {"cycles_per_inst": {"all non-transcendental": CYCLES["slow"], "trans": CYCLES["trans"]}, "source": f"static: {pmc_src}"}
            clk = pmc.get("measured_clock_GHz") or CLOCK_NOMINAL_GHZ
            # the cost table is in cycles of the nominal 2.4 GHz clock it was normalised to: the fraction is against 2.4 GHz x kernel time
            roof["valu_issue_frac"] = {"value": round(cyc / (N_SIMDS * CLOCK_NOMINAL_GHZ * 1e9 * kernel_ms * 1e-3), 4),
                                       "SQ_INSTS_VALU": total, "transcendental_insts": trans, "clock_GHz_in_profiled_run": clk,
                                       "clock_note": "the cost table is TIME per instruction expressed in cycles of 2.4 GHz (measured by tools/experiments/valu_rates.hip on this pool, "
                                                     "where kernels run at 2.0-2.2 GHz): the fraction is issue time / kernel time whatever the clock"}
            roof["valu_issue_frac"].update(pricing)
            if pmc.get("valu_busy_frac") is not None:   # the hardware's own busy counter of the profiled launches (static)
                roof["valu_issue_frac"]["SQ_ACTIVE_INST_VALU_busy_in_profiled_run"] = pmc["valu_busy_frac"]
            roof["hbm_frac"] = round(traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if traffic else None
            busy = max(roof["valu_issue_frac"]["value"], pmc.get("valu_busy_frac") or 0.0)
            roof["bound"] = "valu-issue" if busy > (roof["hbm_frac"] or 0) else "hbm"
            roof["bound_note"] = ("derived: the larger of the VALU issue fraction and hbm_frac; achieved / frac stay the algorithmic-bytes figure "
                                  "against the HBM peak that BASELINE.json's target is stated in")
        roof.update(extras)
        value = args.steps * shaded / dt / 1e6
        result = {
            "metric": "Mshaded-pixels/sec at 4K Sponza, 1 dir + 64 point lights; HBM GB/s vs roofline",
            "value": round(value, 1), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[{args.config - 1}]: {sc.name}, {sc.width}x{sc.height}, 1 dir + {n_lights} point lights, "
                                   f"shadow {sc.shadow_size}^2 PCF 5x5, tonemap {sc.settings[0]}, {len(sc.materials)} materials; shading pass over a resident G-buffer",
                       "triangles": sc.n_triangles, "shaded_pixels": shaded, "sharding": f"{BAND}-row bands round-robin over {world} ranks, RCCL gather to rank 0" if world > 1 else "none",
                       "scale": args.scale, "whole_frame": whole or None,
                       "host_wait": "polling (HSA_ENABLE_INTERRUPT=0)" if os.environ.get("HSA_ENABLE_INTERRUPT") == "0" else "interrupt",
                       "light_culling_rule": "exact only: a pixel skips all lights when fully shadowed (every term of ps_main carries 1 - shadow, forward.hlsl:222,230); "
                                             "the scalar loop also skips a light with n.wi <= 0 in every lit lane of the tile (forward.hlsl:191-192); no range-based "
                                             "tile list: the reference's lights have no range (forward.hlsl:226-230)",
                       "verified_against_single_device_frame": verified,
                       "exchange_path": ("C-ABI arctic_gather_frame (RCCL send/recv + placement kernel)" if cabi else "torch.distributed gather + index_copy_") if world > 1 else None,
                       "exchange": None if gather_ms is None else {
                           "gather_ms": round(gather_ms, 4), "bytes_per_sender": int(pad) * sc.width * 4,
                           "GBps_per_link": round(int(pad) * sc.width * 4 / (gather_ms * 1e-3) / 1e9, 2),
                           "note": "one padded shard per rank into rank 0, timed alone (synchronous), outside the timed steps"}},
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu:
            result["cpu_baseline"] = cpu_baseline(pkg, sc, r, args.cpu_rows)
    for h in rotation[1:]:
        h.close()
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
    # parity in the same run: a stripe of the frame the GPU has just shaded against the float64 oracle
    parity = None
    try:
        r.set_option("keep_float_output", 1)
        r.pass_shade(sc.desc, sc.settings)
        g_ldr, _, g_rgba = r.read_output()
        r.set_option("keep_float_output", 0)
        o.set_precision(64)
        rows_p = min(48, attrs.shape[0])
        y0 = max(0, int(attrs.shape[0] * 0.72) - rows_p // 2)      # through the sunlit floor and its shadow edges
        ref = o.shade_gbuffer(sc.desc, sc.settings, attrs[y0:y0 + rows_p], mat[y0:y0 + rows_p], threads=threads, want=("ldr", "rgba8"))
        err = np.abs(ref["ldr"] - g_ldr[y0:y0 + rows_p])
        parity = {"rows": [y0, y0 + rows_p], "max_abs_ldr_error": float(err.max()), "p9999_ldr_error": float(np.quantile(err, 0.9999)),
                  "rgba8_mismatch_rate": float((ref["rgba8"] != g_rgba[y0:y0 + rows_p]).mean()),
                  "rgba8_max_lsb": int(np.abs(ref["rgba8"].astype(np.int16) - g_rgba[y0:y0 + rows_p].astype(np.int16)).max()),
                  "against": "float64 evaluation of the oracle (oracle/arctic_oracle.cpp) on the same G-buffer, shadow map and lights; gate 1e-4"}
    except Exception as exc:
        parity = {"error": str(exc)}
    o.close()
    model = ""
    try:
        model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except Exception:
        pass
    return {"value": round(px * reps / dt / 1e6, 3), "unit": "Mpixels/s", "cores": threads, "kind": "port",
            "one_thread_value": round(one_thread, 4), "cpu_model": model, "parity_in_this_run": parity,
            "sample": f"rows {start}..{start + n} of the {sc.height}-row frame x {reps} ({px * reps} shaded pixels, {dt:.1f} s), same "
                      f"G-buffer, shadow map and {len(sc.lights)} lights; scalar fp32 C++ oracle, {threads} threads, no culling"}


if __name__ == "__main__":
    main()
