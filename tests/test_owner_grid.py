"""The owner grid of the forward prepass's rasteriser (ARCTIC_OPT_RASTER_OWNER) as plain numbers, on the CPU.

k_raster_owned gives every 16x16 block of a handle's part of the frame ONE owner wave, and the owner's store is also the clear
of the visibility plane: a tile row the grid does not visit keeps last frame's keys, one visited twice is written by two waves
that did not see each other's work items.  The reference draws with the GPU's fixed-function rasteriser over the whole render
target (forward_pass.cpp:212-224), which has no such failure mode, so this invariant is this implementation's own:

    every 8-pixel tile row a shard stores is visited exactly once, as the right one of the shard's packed tile rows,

checked here through the host-side plan (arctic_owner_grid / arctic_owner_visit: the very functions the launch and the kernel
use) for whole frames, row ranges that cut blocks and tiles, and interleaved shards with bands of even and odd numbers of tile
rows, worlds 1..8 -- far more layouts than the GPU tests (tests/test_gpu_raster_owner.py) have time to render."""
import ctypes as C

import numpy as np
import pytest


def plan(lib, width, height, row_begin=0, row_end=0, band_rows=0, shard=(0, 1)):
    grid = (C.c_uint32 * 4)()
    args = (width, height, row_begin, row_end or (0 if band_rows else height), band_rows, shard[0], shard[1])
    assert lib.arctic_owner_grid(*args, grid) == 0
    visits = []
    for gy in range(grid[1]):
        by, rows = C.c_uint32(0), (C.c_int32 * 2)()
        assert lib.arctic_owner_visit(*args, gy, C.byref(by), rows) == 0
        visits.append((by.value, rows[0], rows[1]))
    return tuple(grid), visits


def stored_tile_rows(height, row_begin, row_end, band_rows, shard):
    """global tile rows a shard stores, in the order of its packed local tile rows (restated independently of the library)"""
    all_rows = (height + 7) // 8
    if band_rows:
        bt = band_rows // 8
        return [ty for ty in range(all_rows) if (ty // bt) % shard[1] == shard[0]]
    return list(range(row_begin // 8, (row_end + 7) // 8))


def check(lib, width, height, row_begin=0, row_end=0, band_rows=0, shard=(0, 1)):
    row_end = row_end or height
    grid, visits = plan(lib, width, height, row_begin, 0 if band_rows else row_end, band_rows, shard)
    assert grid[0] == (width + 15) // 16
    want = stored_tile_rows(height, row_begin, row_end, band_rows, shard)
    seen = {}
    blocks = [v[0] for v in visits]
    assert len(set(blocks)) == len(blocks), "a block row is in the grid twice"
    for by, upper, lower in visits:
        for j, local in ((0, upper), (1, lower)):
            if local < 0:
                continue
            assert local not in seen, f"local tile row {local} written by two owners"
            seen[local] = 2 * by + j
    assert sorted(seen) == list(range(len(want))), f"stored tile rows {len(want)}, visited {sorted(seen)}"
    assert [seen[k] for k in range(len(want))] == want      # ... and each as the right global tile row
    return grid


@pytest.fixture(scope="module")
def lib(pkg):
    from arctic_renderer_amd import binding
    return binding.lib()


@pytest.mark.parametrize("size", [(1, 1), (16, 16), (17, 9), (250, 131), (3840, 2160), (7680, 4320), (333, 2161)])
def test_whole_frames(lib, size):
    grid = check(lib, *size)
    assert grid[1] == (size[1] + 15) // 16 and grid[2] == 0


def test_row_ranges(lib):
    rng = np.random.default_rng(3)
    for height in (152, 2160, 1081):
        for _ in range(60):
            a, b = sorted(rng.integers(0, height + 1, 2))
            if a == b:
                continue
            check(lib, 640, height, int(a), int(b))
    for world in range(2, 9):       # the contiguous sharding of sharding.py: equal row counts, not aligned to anything
        edges = [2160 * k // world for k in range(world + 1)]
        for k in range(world):
            check(lib, 3840, 2160, edges[k], edges[k + 1])


@pytest.mark.parametrize("band_rows", [8, 16, 24, 32, 40, 64])
@pytest.mark.parametrize("height", [2160, 4320, 1081, 152, 24])
def test_interleaved_shards(lib, band_rows, height):
    for world in range(1, 9):
        for rank in range(world):
            if rank * band_rows >= height:      # (arctic_create refuses a shard without rows)
                continue
            grid = check(lib, 3840, height, band_rows=band_rows, shard=(rank, world))
            if (band_rows // 8) % 2 == 0:       # the grid holds the shard's own block rows only
                assert grid[3] == 1 and grid[1] <= ((height + 15) // 16 + world - 1) // world + band_rows // 16


def test_bad_arguments(lib):
    grid = (C.c_uint32 * 4)()
    assert lib.arctic_owner_grid(0, 10, 0, 10, 0, 0, 1, grid) != 0
    assert lib.arctic_owner_grid(10, 10, 5, 5, 0, 0, 1, grid) != 0
    assert lib.arctic_owner_grid(10, 10, 0, 11, 0, 0, 1, grid) != 0
    assert lib.arctic_owner_grid(10, 10, 0, 0, 12, 0, 2, grid) != 0      # bands are multiples of 8 rows
    assert lib.arctic_owner_grid(10, 10, 0, 0, 8, 2, 2, grid) != 0
    assert lib.arctic_owner_grid(10, 10, 0, 10, 0, 0, 1, None) != 0
