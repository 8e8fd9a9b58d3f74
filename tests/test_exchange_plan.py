"""The plan of the multi-GPU exchange step (include/arctic_dist.h: arctic_exchange_plan / _row_source / _transfers), on the CPU.

arctic_comm_init, arctic_gather_frame and the placement kernel are built on these three pure host functions, so what the R > 1
branch -- which no one-GPU box can execute -- will post and place is checked here for worlds 2..8 at 2160 and 4320 rows (the 4K
frame has 135 bands of 16 rows: unequal shards at every world size), at heights no band count divides, and for row ranges.
The expected values come from an independent restatement in numpy (sharding.owned_rows / row_range), not from the library."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def sh(pkg):
    from importlib import import_module
    return import_module("arctic_renderer_amd.sharding")


HEIGHTS = (2160, 4320, 2161, 1080, 100, 17, 16, 15, 1)


@pytest.mark.parametrize("world", range(1, 9))
@pytest.mark.parametrize("band", (8, 16, 32))
def test_interleaved_band_plan(sh, world, band):
    for height in HEIGHTS:
        for width in (3840, 7680, 5):
            rows, off, total = sh.exchange_plan(width, height, world, band_rows=band)
            want = [sh.owned_rows(height, k, world, band) for k in range(world)]
            np.testing.assert_array_equal(rows, [len(w) for w in want])
            row_bytes = width * 4
            np.testing.assert_array_equal(off, np.concatenate([[0], np.cumsum(rows[:-1].astype(np.uint64) * row_bytes)]))
            assert total == height * row_bytes                     # the shards tile the staging buffer: no gap, no overlap
            if width != 3840:
                continue
            # what the placement kernel consumes: frame row y <- staging[off[owner] + local * row_bytes]
            seen = np.zeros(total // row_bytes, bool)
            for y in (range(height) if height <= 2161 else sorted(set(range(0, height, 7)) | {height - 1})):
                owner, local = sh.exchange_row_source(y, height, world, band_rows=band)
                assert want[owner][local] == y                   # rank `owner`'s local row `local` IS frame row y
                slot = int(off[owner]) // row_bytes + local
                assert not seen[slot]
                seen[slot] = True
            if height <= 2161:
                assert seen.all()


@pytest.mark.parametrize("world", range(2, 9))
def test_transfers_pair_up_and_tile_the_staging_buffer(sh, world):
    width = 3840
    for height, band in ((2160, 16), (4320, 16), (2161, 16), (100, 8), (17, 16)):
        rows, off, total = sh.exchange_plan(width, height, world, band_rows=band)
        for root in (0, world - 1):
            posts = {k: sh.exchange_transfers(width, world, k, root, rows, off) for k in range(world)}
            sends = {k: p for k, p in posts.items() if k != root}
            recvs = posts[root]
            assert all(not is_send for _, is_send, _, _ in recvs)
            covered = [(int(off[root]), int(off[root]) + int(rows[root]) * width * 4)]      # the root's own shard: a local copy
            for k, p in sends.items():
                if rows[k] == 0:
                    assert p == []                               # an empty shard posts nothing, on either side
                    assert all(peer != k for peer, _, _, _ in recvs)
                    continue
                assert p == [(root, 1, 0, int(rows[k]) * width * 4)]
                match = [t for t in recvs if t[0] == k]
                assert len(match) == 1 and match[0][3] == p[0][3] and match[0][2] == int(off[k])    # same bytes on both sides
                covered.append((match[0][2], match[0][2] + match[0][3]))
            covered = sorted(c for c in covered if c[1] > c[0])
            assert covered[0][0] == 0 and covered[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(covered, covered[1:]))      # disjoint, gap-free


def test_row_range_plan(sh):
    width, height = 3840, 2161
    for world in (1, 2, 3, 8):
        ranges = np.array([sh.row_range(height, k, world) for k in range(world)], np.uint32)
        rows, off, total = sh.exchange_plan(width, height, world, row_ranges=ranges)
        np.testing.assert_array_equal(rows, ranges[:, 1] - ranges[:, 0])
        assert total == height * width * 4
        for y in (0, 1, height // 2, height - 1):
            owner, local = sh.exchange_row_source(y, height, world, row_ranges=ranges)
            assert ranges[owner, 0] <= y < ranges[owner, 1] and local == y - ranges[owner, 0]
    gap = np.array([[0, 10], [20, 30]], np.uint32)                  # rows 10..19 and 30.. belong to nobody
    assert sh.exchange_row_source(15, 40, 2, row_ranges=gap) is None
    assert sh.exchange_row_source(25, 40, 2, row_ranges=gap) == (1, 5)
    with pytest.raises(ValueError):
        sh.exchange_plan(width, height, 2, row_ranges=np.array([[0, 10], [10, height + 1]], np.uint32))
    with pytest.raises(ValueError):
        sh.exchange_plan(width, height, 2)                          # neither bands nor ranges
    with pytest.raises(ValueError):
        sh.exchange_transfers(width, 2, 2, 0, [1, 1], [0, 4])       # rank outside the world


def test_padded_torch_plan_agrees_with_the_c_abi_plan(sh):
    """bench.py's torch.distributed fallback pads every shard to the largest one; its real rows are the C-ABI plan's rows."""
    for height, world, band in ((2160, 8, 16), (2160, 2, 16), (4320, 8, 16), (100, 4, 8)):
        pad, dest = sh.padded_gather_plan(height, world, band)
        rows, _, _ = sh.exchange_plan(3840, height, world, band_rows=band)
        assert pad == rows.max()
        for k in range(world):
            real = dest[k * pad:k * pad + rows[k]]
            for j in (0, int(rows[k]) - 1):
                assert sh.exchange_row_source(int(real[j]), height, world, band_rows=band) == (k, j)
