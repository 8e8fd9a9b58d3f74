"""Driver of tests/test_gpu_exchange_loopback.py (run as a child process, with ARCTIC_RCCL_LIB pointing at the loopback communicator:
the library resolves its nccl* entry points once per process).

R ranks = R threads of this process, each with its own handle on the ONE GPU, driving the C-ABI exchange exactly as bench.py's
ranks do: arctic_comm_init (its layout all-gather), frames rendered into two alternating shard buffers, arctic_gather_frame after
each (the grouped send / recv into the root + the placement kernel, overlapping the next frame), the sharded shadow map's in-place
all-gather when the sun moves.  Checked: the root's assembled frames == the single-device frame, byte for byte.

With `streams` every rank's handle runs on a caller's stream (arctic_set_stream before arctic_comm_init, as bench.py's ranks do): the handle's
own stream is then the exchange stream only, and frames stay three in flight on the caller's stream + the remaining prepass stream.

With `ownback` (implies streams) every rank returns to the handle's own stream (arctic_use_own_stream) between frames 1 and 2, with gather 1 in flight
on the stream the exchange had borrowed: the exchange keeps that stream, the passes move to a fresh one.

usage: python tests/loopback_worlds.py <world> <bands|rows> [shadow] [streams] [ownback]     prints LOOPBACK_OK on success"""
import os
import sys
import threading
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    world, layout = int(sys.argv[1]), sys.argv[2]
    sharded_shadow = "shadow" in sys.argv[3:]
    own_back = "ownback" in sys.argv[3:]
    callers_streams = "streams" in sys.argv[3:] or own_back
    assert os.environ.get("ARCTIC_RCCL_LIB"), "the loopback communicator is selected with ARCTIC_RCCL_LIB"
    import torch
    pkg = entry.load_package()
    hip = pkg.renderer
    sc = pkg.scenes.config3(scale=0.13)
    full = sc.upload(hip.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
    want = full.render_frame(sc.desc, sc.settings).copy()
    full.close()

    band = 16
    if layout == "rows":      # contiguous row ranges of unequal size; from three ranks on the first shard is ONE row
        if world >= 3:
            inner = [1] + [1 + (sc.height - 1) * k // (world - 1) + (5 * k) % 11 for k in range(1, world - 1)]
        else:
            inner = [sc.height // 2 + 3]
        cuts = [0] + inner + [sc.height]
        assert len(cuts) == world + 1 and all(a < b for a, b in zip(cuts[:-1], cuts[1:]))
    uid = hip.Renderer.comm_unique_id()
    frames = [torch.zeros((sc.height, sc.width, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
    errors, results = [], {}
    start = threading.Barrier(world)

    def rank_main(rank):
        try:
            if layout == "bands":
                r = sc.upload(hip.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights, band_rows=band, shard=(rank, world)))
            else:
                r = sc.upload(hip.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights, row_begin=cuts[rank], row_end=cuts[rank + 1]))
            stream = None
            if callers_streams:
                stream = torch.cuda.Stream()
                r.set_stream(stream.cuda_stream)
            start.wait()
            r.comm_init(uid, rank, world)
            if sharded_shadow:
                r.set_option("shadow_sharded", 1)
                r.set_option("shadow_cache", 0)        # the map is redrawn -- in row slices, all-gathered -- every frame
            outs = [torch.zeros((max(r.rows, 1), sc.width, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
            torch.cuda.synchronize()                   # (the fills run on torch's stream, the frames on the handle's)
            for k in range(4):                         # alternating shard buffers: gather k overlaps frame k + 1
                b = k % 2
                if own_back and k == 2:
                    r.set_stream(None)                 # arctic_use_own_stream, gather 1 possibly unmatched on the borrowed stream
                r.render_frame_device(sc.desc, sc.settings, outs[b].data_ptr())
                r.gather_frame(outs[b].data_ptr(), frames[b].data_ptr() if rank == 0 else None, 0)
            r.flush()
            start.wait()                               # every rank's transfers are complete before anyone tears down
            if rank == 0:
                results["frames"] = [f.cpu().numpy() for f in frames]
            if sharded_shadow:
                results[("map", rank)] = r.read_shadow_map()
            start.wait()
            r.comm_destroy()
            r.close()
        except Exception:
            errors.append(f"rank {rank}: {traceback.format_exc()}")
            try:
                start.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=rank_main, args=(k,)) for k in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=240)
    if errors or any(t.is_alive() for t in threads):
        print("\n".join(errors) or "a rank thread did not finish", flush=True)
        os._exit(1)
    for f in results["frames"]:
        np.testing.assert_array_equal(f, want)
    if sharded_shadow:
        single = sc.upload(hip.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
        single.pass_shadow_map(sc.desc)
        ref_map = single.read_shadow_map()
        single.close()
        for rank in range(world):
            np.testing.assert_array_equal(results[("map", rank)].view(np.uint32), ref_map.view(np.uint32))
    print(f"LOOPBACK_OK world {world} {layout}{' sharded-shadow' if sharded_shadow else ''}{' callers-streams' if callers_streams else ''}{' own-stream-again' if own_back else ''}", flush=True)


if __name__ == "__main__":
    main()
