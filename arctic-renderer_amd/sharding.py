"""Screen-space sharding of one frame across the GPUs of a node.

The reference is single-GPU (src/renderer/rhi.cpp:120-124); this is the MI355X-side extension
BASELINE.json's north_star asks for.  Every pixel of the shading pass reads only its own G-buffer
texel plus read-only scene data (forward.hlsl:208-235), so rank r of R owns the contiguous rows
[r*H/R, (r+1)*H/R): no halo, no data-path collective while shading, and ONE exchange step at the
end -- the gather of the finished RGBA8 shards on rank 0.  On the xGMI full mesh that gather is R-1
independent point-to-point transfers, one per link into the root.

One process per GPU, torch.distributed (backend "nccl" = RCCL on ROCm; "gloo" on CPU for tests).
"""
import torch
import torch.distributed as dist


def row_range(height, rank, world):
    """rows [begin, end) of rank `rank`; balanced to within one row, contiguous, covering [0, height)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    base, extra = divmod(height, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def owned_rows(height, rank, world, band_rows=16):
    """row indices of the INTERLEAVED shard of rank `rank`: bands of band_rows rows dealt round-robin (band b -> rank
    b % world).  Lit (expensive) regions are spatially clustered, so contiguous row ranges would leave some ranks with all
    the light evaluations; interleaving gives every rank its share."""
    import numpy as np
    if band_rows <= 0 or band_rows % 8:
        raise ValueError("band_rows must be a positive multiple of 8")
    y = np.arange(height)
    return y[(y // band_rows) % world == rank]


def exchange_plan(width, height, world, band_rows=0, row_ranges=None):
    """arctic_exchange_plan (include/arctic_dist.h; a pure host function of libarctic_hip.so, no device needed): rows of every
    rank's shard, the byte offset of every shard in the root's staging buffer, and that buffer's size."""
    import ctypes as C
    import numpy as np
    from . import binding
    rows, off, total = np.zeros(world, np.uint32), np.zeros(world, np.uint64), C.c_uint64(0)
    rr = None if row_ranges is None else np.ascontiguousarray(row_ranges, dtype=np.uint32).reshape(-1)
    rc = binding.lib().arctic_exchange_plan(width, height, band_rows, world, None if rr is None else rr.ctypes.data_as(C.c_void_p),
                                            rows.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p), C.byref(total))
    if rc != 0:
        raise ValueError(f"arctic_exchange_plan: {binding.ERRORS.get(rc, rc)}")
    return rows, off, int(total.value)


def exchange_transfers(width, world, rank, root, rows, offset):
    """arctic_exchange_transfers: the point-to-point transfers `rank` posts in one arctic_gather_frame, as (peer, is_send,
    staging_offset, bytes) tuples."""
    import ctypes as C
    import numpy as np
    from . import binding
    out = (binding.CTransfer * world)()
    rows, offset = np.ascontiguousarray(rows, np.uint32), np.ascontiguousarray(offset, np.uint64)
    n = binding.lib().arctic_exchange_transfers(width, world, rank, root, rows.ctypes.data_as(C.c_void_p), offset.ctypes.data_as(C.c_void_p),
                                                C.cast(out, C.c_void_p), world)
    if n < 0:
        raise ValueError(f"arctic_exchange_transfers: {binding.ERRORS.get(n, n)}")
    return [(t.peer, t.is_send, t.staging_offset, t.bytes) for t in out[:n]]


def exchange_row_source(y, height, world, band_rows=0, row_ranges=None):
    """arctic_exchange_row_source: (owner, local_row) of frame row y, or None when no rank owns it."""
    import ctypes as C
    import numpy as np
    from . import binding
    rr = None if row_ranges is None else np.ascontiguousarray(row_ranges, dtype=np.uint32).reshape(-1)
    owner, local = C.c_uint32(0), C.c_uint32(0)
    rc = binding.lib().arctic_exchange_row_source(y, height, band_rows, world, None if rr is None else rr.ctypes.data_as(C.c_void_p),
                                                  C.byref(owner), C.byref(local))
    if rc < 0:
        raise ValueError(f"arctic_exchange_row_source: {binding.ERRORS.get(rc, rc)}")
    return None if rc == 1 else (owner.value, local.value)


def padded_gather_plan(height, world, band_rows=16):
    """equal-size exchange for interleaved shards: every rank sends `pad` rows (its own rows first, the rest unused), so
    the gather is ONE collective (ncclGather) whatever the band count.  Returns (pad, dest) where dest[k * pad + j] is the
    frame row of rank k's j-th sent row; the unused tail rows of the shards go to distinct dummy rows height .. world*pad - 1
    of an extended frame (world * pad rows), so the root de-interleaves with one indexed copy of the whole staging buffer."""
    import numpy as np
    rows = [owned_rows(height, k, world, band_rows) for k in range(world)]
    pad = max(len(x) for x in rows)
    dest = np.empty(world * pad, np.int64)
    dummy = height
    for k, x in enumerate(rows):
        dest[k * pad:k * pad + len(x)] = x
        dest[k * pad + len(x):(k + 1) * pad] = dummy + np.arange(pad - len(x))
        dummy += pad - len(x)
    return pad, dest


def assemble_banded(gathered, height, band_rows=16):
    """root only: the full frame from interleaved shards (one indexed copy per rank)."""
    world = len(gathered)
    frame = torch.empty((height,) + tuple(gathered[0].shape[1:]), dtype=gathered[0].dtype, device=gathered[0].device)
    for k, g in enumerate(gathered):
        idx = torch.as_tensor(owned_rows(height, k, world, band_rows), device=g.device)
        frame.index_copy_(0, idx, g)
    return frame


class _Works:
    """several point-to-point requests waited on as one."""

    def __init__(self, reqs):
        self.reqs = reqs

    def wait(self):
        for q in self.reqs:
            q.wait()


def gather_rows(shard, gathered, rank, world, root=0, group=None, async_op=False, equal_rows=None):
    """gather the (rows_r, width, 4) uint8 shards on `root`; `gathered` is the root's list of per-rank buffers (None
    elsewhere).  Equal shards: one dist.gather (ncclGather); shards differing by a row: send/recv into the root.
    async_op=True returns an object with .wait() (None when there is nothing to wait for).
    equal_rows: whether every rank's shard has the same number of rows -- something all ranks must agree on, because it selects
    the collective.  Pass it when the caller knows (bench.py pads every shard to one size: True); None decides with one small
    all_reduce per call (never cached: a cached answer keyed on the local shape can differ between ranks after a resize)."""
    if world == 1:
        return None if async_op else shard
    if shard.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal of the multi-rank path on a box with one GPU (gloo has no device gather): stage through the host
        torch.cuda.current_stream().synchronize()
        host = [torch.empty(g.shape, dtype=g.dtype) for g in gathered] if rank == root else None
        gather_rows(shard.cpu(), host, rank, world, root, group, equal_rows=equal_rows)
        if rank == root:
            for g, h in zip(gathered, host):
                g.copy_(h)
        return _Works([]) if async_op else gathered
    if _all_equal_rows(shard, world, group) if equal_rows is None else equal_rows:   # every rank must take the same branch
        w = dist.gather(shard, gather_list=gathered if rank == root else None, dst=root, group=group, async_op=async_op)
        return w if async_op else gathered
    if rank == root:
        gathered[root].copy_(shard)
        reqs = [dist.irecv(gathered[k], src=k, group=group) for k in range(world) if k != root]
    else:
        reqs = [dist.isend(shard, dst=root, group=group)]
    works = _Works(reqs)
    if async_op:
        return works
    works.wait()
    return gathered


def _all_equal_rows(shard, world, group):
    t = torch.tensor([shard.shape[0], -shard.shape[0]], dtype=torch.int64, device=shard.device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return bool(t[0].item() == -t[1].item())


def assemble(gathered):
    """root only: the full frame (height, width, 4) from the per-rank shards."""
    return torch.cat(gathered, dim=0)
