"""The shadow min/max table of shade.hip (k_shadow_blocks / k_shadow_bounds / shadow_prepare) decides a pixel without running the
25 PCF taps.  Its exactness rests on two claims that can be checked on the CPU against the oracle's literal calculate_shadow
(forward.hlsl:68-96), with the table and the entry selection restated in numpy fp32 exactly as the kernels compute them:

  1. coverage: when the first texel (bx, by) of tap 0 satisfies 0 <= bx, by < S - 3 (and S <= 4900), every texel any of the 25
     bilinear taps reads lies in [bx, bx + 3] x [by, by + 3], inside entry (bx >> 2, by >> 2) = texels [4i, 4i + 8) x [4j, 4j + 8);
  2. interval: a bilinear tap (fmaf lerps with weights in [0, 1)) cannot leave [min, max] of the texels it reads, so
     pz > max  =>  shadow == 1 exactly   and   pz <= min  =>  shadow == 0 exactly.
No GPU involved: this is the host-side proof obligation of the device shortcut.
"""
import ctypes as C

import numpy as np
import pytest

f32 = np.float32


def table(m):
    """min/max over texels [4i, 4i+8) x [4j, 4j+8), clamped at the map's edge: blocks of 4x4 first, then 2x2 blocks of blocks."""
    S = m.shape[0]
    nb = (S + 3) // 4
    pad = nb * 4 - S
    lo = np.pad(m, ((0, pad), (0, pad)), constant_values=np.inf).reshape(nb, 4, nb, 4).min(axis=(1, 3))
    hi = np.pad(m, ((0, pad), (0, pad)), constant_values=-np.inf).reshape(nb, 4, nb, 4).max(axis=(1, 3))
    i1 = np.minimum(np.arange(nb) + 1, nb - 1)
    lo2 = np.minimum(np.minimum(lo, lo[:, i1]), np.minimum(lo[i1, :], lo[i1][:, i1]))
    hi2 = np.maximum(np.maximum(hi, hi[:, i1]), np.maximum(hi[i1, :], hi[i1][:, i1]))
    return lo2, hi2


def first_texel(p, S):
    """floor((p + -0.0002f) * Sf - 0.5f) in fp32, operation by operation (shadow_prepare)"""
    return np.floor((p + f32(-0.0002)) * f32(S) - f32(0.5)).astype(np.int64)


def tap_texels(p, S):
    """x0 of the five taps along one axis, as the oracle's wrap_axis computes them for coordinates inside [0, 1)"""
    out = []
    for i in range(-2, 3):
        u = p + f32(i) * f32(0.0001)
        out.append(np.floor(u * f32(S) - f32(0.5)).astype(np.int64))
    return np.stack(out)


@pytest.mark.parametrize("S", [8, 64, 400, 2048, 4000, 4096, 4900])
def test_footprint_stays_inside_the_window(S):
    rng = np.random.default_rng(S)
    p = rng.random(400000).astype(f32)
    # plus values hugging texel boundaries, where floor() flips
    k = rng.integers(0, S, 100000)
    edge = ((k + 0.5 + rng.choice([-1e-4, 0.0, 1e-4], 100000)) / S + rng.choice([-2e-4, 0.0, 2e-4], 100000)).astype(f32)
    p = np.concatenate([p, edge])
    b = first_texel(p, S)
    ok = (b >= 0) & (b < S - 3)
    t = tap_texels(p[ok], S)
    assert (t.min(axis=0) >= b[ok]).all()
    assert (t.max(axis=0) + 1 <= b[ok] + 3).all(), "a tap reads a texel outside the 4-wide window behind the first texel"
    # and the window is inside the table entry of its first texel
    assert ((b[ok] >> 2) * 4 <= b[ok]).all() and (b[ok] + 3 < (b[ok] >> 2) * 4 + 8).all()
    # no tap wraps: all tap coordinates are inside [0, 1)
    u_lo, u_hi = p[ok] + f32(-2) * f32(0.0001), p[ok] + f32(2) * f32(0.0001)
    assert (u_lo >= 0).all() and (u_hi < 1).all()


def maps(rng, S):
    yield "noise", (0.3 + 0.5 * rng.random((S, S))).astype(f32)
    y, x = np.mgrid[0:S, 0:S]
    yield "edges", np.where((x // 7 + y // 5) % 2 == 0, 0.35, 0.8).astype(f32) + (rng.random((S, S)) * 1e-3).astype(f32)
    yield "ramp", (0.2 + 0.6 * (x + 2 * y) / (3.0 * S)).astype(f32)
    m = np.ones((S, S), f32); m[S // 4: S // 2, S // 3: 2 * S // 3] = f32(0.5)
    yield "cleared map with one occluder", m


@pytest.mark.parametrize("S", [64, 400, 1000])
def test_table_decisions_equal_the_25_tap_result(oracle, S):
    L = oracle.lib()
    L.oracle_calculate_shadow.restype = C.c_float
    L.oracle_calculate_shadow.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    rng = np.random.default_rng(1000 + S)
    decided = 0
    for name, m in maps(rng, S):
        lo, hi = table(m)
        n = 6000
        px = rng.random(n).astype(f32); py = rng.random(n).astype(f32)
        bx, by = first_texel(px, S), first_texel(py, S)
        # depths: random, and right at the entry's bounds (the comparisons are strict / non-strict)
        i, j = np.clip(bx >> 2, 0, lo.shape[1] - 1), np.clip(by >> 2, 0, lo.shape[0] - 1)
        pick = rng.integers(0, 5, n)
        pz = np.select([pick == 0, pick == 1, pick == 2, pick == 3], [lo[j, i], hi[j, i], np.nextafter(hi[j, i], f32(2)), np.nextafter(lo[j, i], f32(-1))],
                       rng.random(n).astype(f32)).astype(f32)
        for k in range(n):
            if not (0 <= bx[k] < S - 3 and 0 <= by[k] < S - 3) or pz[k] > 1.0:
                continue
            # light-space position that maps to (px, py, pz): px = x * 0.5 + 0.5, py = 1 - (y * 0.5 + 0.5) -- solved in float64, then
            # checked to round-trip in fp32 (otherwise skip: the device sees the fp32 values)
            ls = np.array([2.0 * float(px[k]) - 1.0, 1.0 - 2.0 * float(py[k]), float(pz[k]), 1.0], f32)
            rx = ls[0] * f32(0.5) + f32(0.5); ry = f32(1.0) - (ls[1] * f32(0.5) + f32(0.5))
            if first_texel(np.array([rx]), S)[0] != bx[k] or first_texel(np.array([ry]), S)[0] != by[k]:
                continue
            got = L.oracle_calculate_shadow(m.ctypes.data, S, ls.ctypes.data)
            if pz[k] > hi[j[k], i[k]]:
                assert got == 1.0, (name, k, got)
                decided += 1
            elif not (pz[k] > lo[j[k], i[k]]):
                assert got == 0.0, (name, k, got)
                decided += 1
    assert decided > 2000
