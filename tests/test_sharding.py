"""Row sharding + gather of the multi-GPU path, rehearsed on CPU with gloo (world_size 2 and 3).

The shards are produced by the CPU oracle here (no GPU in this container); the GPU version of the same invariant
(sharded frame == single-device frame, bit for bit) is tests/test_gpu_parity.py::test_row_shards_equal_full_frame."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_row_range_partitions_the_frame(pkg):
    from importlib import import_module
    sh = import_module("arctic_renderer_amd.sharding")
    for h in (1, 7, 8, 270, 1080, 2160, 4320, 2161):
        for w in (1, 2, 3, 4, 8):
            if w > h:
                continue
            rows = [sh.row_range(h, r, w) for r in range(w)]
            assert rows[0][0] == 0 and rows[-1][1] == h
            assert all(a[1] == b[0] for a, b in zip(rows, rows[1:]))
            sizes = [b - a for a, b in rows]
            assert max(sizes) - min(sizes) <= 1
    assert sh.row_range(2160, 3, 8) == (810, 1080) and sh.row_range(4320, 7, 8) == (3780, 4320)
    with pytest.raises(ValueError):
        sh.row_range(100, 4, 4)


def test_owned_rows_partition_and_assemble(pkg):
    from importlib import import_module
    sh = import_module("arctic_renderer_amd.sharding")
    for h, world, band in ((2160, 8, 16), (216, 3, 16), (100, 4, 8), (4320, 8, 32)):
        rows = [sh.owned_rows(h, r, world, band) for r in range(world)]
        allr = np.sort(np.concatenate(rows))
        np.testing.assert_array_equal(allr, np.arange(h))                     # a partition of the frame
        assert max(map(len, rows)) - min(map(len, rows)) <= band
        frame = torch.arange(h * 5 * 4, dtype=torch.int64).reshape(h, 5, 4).to(torch.uint8)
        parts = [frame[torch.as_tensor(r)] for r in rows]
        assert torch.equal(sh.assemble_banded(parts, h, band), frame)
    with pytest.raises(ValueError):
        sh.owned_rows(100, 0, 2, 12)


def test_padded_gather_plan_is_one_indexed_copy(pkg):
    """equal-size exchange (what bench.py uses for N > 1): staging row k*pad+j -> frame row dest[k*pad+j]."""
    from importlib import import_module
    sh = import_module("arctic_renderer_amd.sharding")
    for h, world, band in ((2160, 8, 16), (2160, 2, 16), (2160, 4, 16), (216, 3, 16), (100, 4, 8), (4320, 8, 16)):
        pad, dest = sh.padded_gather_plan(h, world, band)
        rows = [sh.owned_rows(h, r, world, band) for r in range(world)]
        assert pad == max(map(len, rows)) and len(dest) == world * pad
        real = dest[dest < h]
        np.testing.assert_array_equal(np.sort(real), np.arange(h))          # every frame row exactly once
        np.testing.assert_array_equal(np.sort(dest), np.arange(world * pad))  # a permutation: padding rows land on distinct dummy rows
        frame = torch.arange(h * 3 * 4, dtype=torch.int64).reshape(h, 3, 4).to(torch.uint8)
        staging = torch.full((world * pad, 3, 4), 77, dtype=torch.uint8)
        for k, x in enumerate(rows):
            staging[k * pad:k * pad + len(x)] = frame[torch.as_tensor(x)]
        ext = torch.zeros((world * pad, 3, 4), dtype=torch.uint8)
        ext.index_copy_(0, torch.as_tensor(dest), staging)
        assert torch.equal(ext[:h], frame)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, height_trim, out_path, use_async=False):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry
    from importlib import import_module
    from oracle import oracle as O
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pkg = entry.load_package()
        sh = import_module("arctic_renderer_amd.sharding")
        sc = pkg.scenes.config3(scale=0.04, tex=64)
        height = sc.height - height_trim            # an odd height makes the shards unequal (send/recv path)
        sc.desc.camera["aspect"] = sc.width / height
        b, e = sh.row_range(height, rank, world)
        o = O.Oracle(sc.width, height, sc.shadow_size, sc.max_lights, row_begin=b, row_end=e)
        sc.upload(o)
        shard = torch.from_numpy(o.render_frame(sc.desc, sc.settings, threads=2))
        gathered = [torch.empty((sh.row_range(height, k, world)[1] - sh.row_range(height, k, world)[0], sc.width, 4), dtype=torch.uint8)
                    for k in range(world)] if rank == 0 else None
        if use_async:
            w = sh.gather_rows(shard, gathered, rank, world, async_op=True)
            w.wait()
        else:
            sh.gather_rows(shard, gathered, rank, world)
        if rank == 0:
            np.save(out_path, sh.assemble(gathered).numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,trim,use_async", [(2, 0, False), (2, 3, False), (3, 1, True), (2, 0, True)],
                         ids=["2-equal", "2-ragged", "3-ragged-async", "2-equal-async"])
def test_gloo_gather_reassembles_the_single_process_frame(pkg, oracle, tmp_path, world, trim, use_async):
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), trim, out, use_async), nprocs=world, join=True)
    got = np.load(out)
    sc = pkg.scenes.config3(scale=0.04, tex=64)
    height = sc.height - trim
    sc.desc.camera["aspect"] = sc.width / height
    ref = sc.upload(oracle.Oracle(sc.width, height, sc.shadow_size, sc.max_lights)).render_frame(sc.desc, sc.settings, threads=4)
    assert got.shape == ref.shape
    np.testing.assert_array_equal(got, ref)      # sharding must not change a single byte


def _band_worker(rank, world, port, band, padded, out_path):
    """interleaved bands (what bench.py shards with): every rank shades the rows it owns -- here the oracle's frame cut to those
    rows -- and the root puts the frame together, either bench.py's way (every shard padded to the largest: ONE equal-count gather,
    one indexed copy through padded_gather_plan) or shard by shard (unequal send/recv + assemble_banded).  The row bookkeeping of both
    is cross-checked against the C-ABI plan (arctic_exchange_plan), which is what the RCCL path places rows by."""
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry
    from importlib import import_module
    from oracle import oracle as O
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pkg = entry.load_package()
        sh = import_module("arctic_renderer_amd.sharding")
        sc = pkg.scenes.config3(scale=0.04, tex=64)
        o = sc.upload(O.Oracle(sc.width, sc.height, sc.shadow_size, sc.max_lights))
        mine = sh.owned_rows(sc.height, rank, world, band)
        shard = torch.from_numpy(o.render_frame(sc.desc, sc.settings, threads=2)[mine])
        rows, off, total = sh.exchange_plan(sc.width, sc.height, world, band_rows=band)
        assert rows[rank] == len(mine) and total == sc.height * sc.width * 4
        if padded:
            pad, dest = sh.padded_gather_plan(sc.height, world, band)
            out = torch.zeros((pad, sc.width, 4), dtype=torch.uint8)
            out[:len(mine)] = shard
            staging = torch.zeros((world * pad, sc.width, 4), dtype=torch.uint8) if rank == 0 else None
            gathered = [staging[k * pad:(k + 1) * pad] for k in range(world)] if rank == 0 else None
            w = sh.gather_rows(out, gathered, rank, world, async_op=True, equal_rows=True)
            w.wait()
            if rank == 0:
                ext = torch.zeros((world * pad, sc.width, 4), dtype=torch.uint8)
                ext.index_copy_(0, torch.as_tensor(dest), staging)
                np.save(out_path, ext[:sc.height].numpy())
        else:
            gathered = [torch.empty((int(rows[k]), sc.width, 4), dtype=torch.uint8) for k in range(world)] if rank == 0 else None
            sh.gather_rows(shard, gathered, rank, world)        # unequal shards: decided by one all_reduce, then send/recv
            if rank == 0:
                np.save(out_path, sh.assemble_banded(gathered, sc.height, band).numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,band,padded", [(2, 16, True), (3, 8, True), (3, 8, False), (2, 8, False)],
                         ids=["2x16-padded", "3x8-padded", "3x8-sendrecv", "2x8-sendrecv"])
def test_gloo_interleaved_bands_reassemble_the_frame(pkg, oracle, tmp_path, world, band, padded):
    out = str(tmp_path / "frame.npy")
    mp.spawn(_band_worker, args=(world, _free_port(), band, padded, out), nprocs=world, join=True)
    sc = pkg.scenes.config3(scale=0.04, tex=64)
    ref = sc.upload(oracle.Oracle(sc.width, sc.height, sc.shadow_size, sc.max_lights)).render_frame(sc.desc, sc.settings, threads=4)
    np.testing.assert_array_equal(np.load(out), ref)
