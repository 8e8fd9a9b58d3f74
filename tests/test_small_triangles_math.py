"""The integer identities the shadow pass's small-triangle path rests on (arctic-renderer_amd/csrc/geometry.hip: SmallRec, stage_small, draw_small), restated in
numpy and checked exhaustively / on random inputs on the CPU.  (The device code itself runs against the oracle in tests/test_gpu_small_triangles.py.)

Reference: the hardware rasteriser the pass stands in for, src/renderer/shadow_map_pass.cpp:157-167 -- nothing of this arithmetic exists there."""
import numpy as np


def test_slot_to_box_coordinates_is_exact():
    """slot j of a box of width w is pixel (j mod w, j div w); the kernel computes j div w as (j * M) >> 16 with M = floor(65536 / w) + 1 in 24-bit multiplies.
    Exact for every width it admits (w <= 64) and every slot of a box (j < w h <= SMALL_PX = 64; also for the 256 of the measured variant), and more generally
    while w * j < 65536."""
    for w in range(1, 65):
        M = 65536 // w + 1
        assert M < (1 << 24)
        j = np.arange(0, 256, dtype=np.int64)
        assert (j * M < (1 << 32)).all()
        np.testing.assert_array_equal((j * M) >> 16, j // w)
        jj = np.arange(0, (65536 + w - 1) // w, dtype=np.int64)      # the general bound: w * j < 65536
        np.testing.assert_array_equal((jj * M) >> 16, jj // w)
    w, M = 64, 65536 // 64 + 1                                       # ... and it IS the bound: the first slot beyond it that breaks
    j = np.arange(0, 1 << 17, dtype=np.int64)
    bad = np.nonzero(((j * M) >> 16) != j // w)[0]
    assert bad.size and bad[0] * w >= 65536


def test_start_bits_name_the_triangle_of_every_slot():
    """a wave's small triangles get runs of pixel slots by a prefix sum; a lane finds the triangle of slot p from the 64 start bits of its step:
    triangles begun in earlier steps + start bits at or below its lane - 1 (draw_small)."""
    rng = np.random.default_rng(7)
    for trial in range(200):
        k = int(rng.integers(1, 65))                                  # small triangles among the wave's 64
        n = rng.integers(1, 65, k)                                    # 1 .. SMALL_PX slots each
        first = np.concatenate([[0], np.cumsum(n)[:-1]])
        total = int(n.sum())
        bits = np.zeros((total + 63) // 64 * 64 + 64, dtype=bool)
        bits[first] = True
        expect = np.repeat(np.arange(k), n)
        base = 0
        for step in range(0, total, 64):
            word = bits[step:step + 64]
            lanes = np.arange(64)
            idx = base + np.cumsum(word) - 1                          # popcount(start bits & lanes <= lane) - 1
            base += int(word.sum())
            live = step + lanes < total
            np.testing.assert_array_equal(idx[live], expect[step:step + 64][:live.sum()])


def test_edge_functions_relative_to_the_box_stay_in_int32():
    """small_record refuses a triangle unless |step| < 2^23 for both axes and |E| + |Ax| (w - 1) + |By| (h - 1) < 2^31 - 16: then every value E + Ax dx + By dy over
    the box is an int32 and both factors of each product fit the 24-bit multiplier (dx, dy < 64)."""
    rng = np.random.default_rng(11)
    for trial in range(2000):
        w, h = int(rng.integers(1, 65)), int(rng.integers(1, 65))
        if w * h > 64:
            continue
        ax, by = (int(rng.integers(-(1 << 24), 1 << 24)) for _ in range(2))
        E = int(rng.integers(-(1 << 33), 1 << 33))
        fits = abs(ax) < (1 << 23) and abs(by) < (1 << 23) and abs(E) + abs(ax) * (w - 1) + abs(by) * (h - 1) < 0x7FFFFFF0
        if not fits:
            continue
        dx, dy = np.meshgrid(np.arange(w), np.arange(h))
        v = E + ax * dx.astype(object) + by * dy.astype(object)
        assert all(-(1 << 31) <= int(x) < (1 << 31) for x in v.ravel())
        assert -(1 << 23) <= ax < (1 << 23) and -(1 << 23) <= by < (1 << 23)
