"""The multi-GPU exchange steps of include/arctic_dist.h on the one GPU a test box has.

What can run here: (1) the placement kernel (the root's half of arctic_gather_frame) with the shards of several handles that share
one GPU -- interleaved bands and row ranges -- against the single-device frame, byte for byte; (2) a ONE-rank RCCL communicator
through the whole API: ncclGetUniqueId / ncclCommInitRank, the layout all-gather, arctic_gather_frame (overlapped, double
buffered), the sharded shadow map (a one-rank all-gather), destroy.  The R > 1 transfers themselves (grouped ncclSend / ncclRecv)
need R GPUs: RCCL refuses two ranks on one device; they run in the driver's scaling bench (bench.py --gpus N).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scene(pkg):
    return pkg.scenes.config3(scale=0.12)


@pytest.fixture(scope="module")
def full_frame(scene, hip):
    r = scene.upload(hip.Renderer(scene.width, scene.height, scene.shadow_size, scene.max_lights))
    img = r.render_frame(scene.desc, scene.settings)
    r.close()
    return img


def dev(a):
    import torch
    return torch.as_tensor(a, device="cuda")


@pytest.mark.parametrize("world,band", [(2, 16), (3, 8), (5, 16)])
def test_assemble_interleaved_bands(scene, hip, full_frame, world, band):
    import torch
    sc = scene
    shards, handles = [], []
    for k in range(world):
        r = sc.upload(hip.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights, band_rows=band, shard=(k, world)))
        shards.append(r.render_frame(sc.desc, sc.settings))
        handles.append(r)
    staging = dev(np.concatenate([s.reshape(-1) for s in shards]))
    for k in (0, world - 1):   # any handle of the sharding can place
        frame = torch.zeros((sc.height, sc.width, 4), dtype=torch.uint8, device="cuda")
        handles[k].assemble_frame(staging.data_ptr(), frame.data_ptr(), world)
        handles[k].flush()
        np.testing.assert_array_equal(frame.cpu().numpy(), full_frame)
    for r in handles:
        r.close()


def test_assemble_row_ranges(scene, hip, full_frame):
    import torch
    sc = scene
    cuts = [0, 37, 38, sc.height - 50, sc.height]       # unequal shards, one of a single row
    shards = []
    for b, e in zip(cuts[:-1], cuts[1:]):
        r = sc.upload(hip.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights, row_begin=b, row_end=e))
        shards.append(r.render_frame(sc.desc, sc.settings))
        r.close()
    r = sc.upload(hip.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights))
    staging = dev(np.concatenate([s.reshape(-1) for s in shards]))
    frame = torch.zeros((sc.height, sc.width, 4), dtype=torch.uint8, device="cuda")
    ranges = np.array([[b, e] for b, e in zip(cuts[:-1], cuts[1:])], np.uint32)
    r.assemble_frame(staging.data_ptr(), frame.data_ptr(), len(ranges), ranges)
    r.flush()
    np.testing.assert_array_equal(frame.cpu().numpy(), full_frame)
    with pytest.raises(hip.ArcticError):
        r.assemble_frame(staging.data_ptr(), frame.data_ptr(), 2, np.array([[0, 10], [10, sc.height + 1]], np.uint32))
    r.close()


@pytest.mark.parametrize("callers_stream", [False, True], ids=["own-stream", "callers-stream"])
def test_one_rank_communicator_end_to_end(scene, hip, full_frame, callers_stream):
    """(callers-stream: the host brings its stream, as bench.py does: the exchange then runs on the handle's own, idle stream)"""
    import torch
    sc = scene
    r = sc.upload(hip.Renderer(sc.width, sc.height, sc.shadow_size, sc.max_lights, band_rows=16, shard=(0, 1)))
    if callers_stream:
        r.set_stream(torch.cuda.current_stream().cuda_stream)
    with pytest.raises(hip.ArcticError):
        r.gather_frame(None, 1, 0)                       # no communicator yet
    uid = hip.Renderer.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    r.comm_init(uid, 0, 1)
    with pytest.raises(hip.ArcticError):
        r.comm_init(uid, 0, 1)                           # one communicator per handle
    r.set_option("shadow_sharded", 1)                    # world 1: the whole map is this rank's slice
    outs = [torch.zeros((sc.height, sc.width, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
    frames = [torch.zeros((sc.height, sc.width, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
    for k in range(6):                                   # alternating shard buffers: gather k overlaps frame k + 1
        b = k % 2
        r.render_frame_device(sc.desc, sc.settings, outs[b].data_ptr())
        r.gather_frame(outs[b].data_ptr(), frames[b].data_ptr(), 0)
    r.flush()
    for f in frames:
        np.testing.assert_array_equal(f.cpu().numpy(), full_frame)
    r.render_frame(sc.desc, sc.settings)                 # the handle's own output buffer as the shard
    frame = torch.zeros((sc.height, sc.width, 4), dtype=torch.uint8, device="cuda")
    r.gather_frame(None, frame.data_ptr(), 0)
    r.flush()
    np.testing.assert_array_equal(frame.cpu().numpy(), full_frame)
    r.comm_destroy()
    r.comm_init(hip.Renderer.comm_unique_id(), 0, 1)     # and again after a destroy
    r.close()                                            # close destroys the communicator
