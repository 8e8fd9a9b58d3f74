// shadow_coords.h -- light-space position -> shadow-map coordinates and the entry of the min/max table that covers a pixel's
// 25 PCF taps: ONE definition for the shading kernels (shade.hip: the test itself) and the G-buffer prepass (geometry.hip: the
// per-tile cost hint, "can a pixel of this tile be lit at all?").  Device code only.
#pragma once
#include "common.h"

namespace arctic {

// floor and convert in one instruction (exact for |x| < 2^31)
__device__ __forceinline__ int floor_to_int(float x) { int i; asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(i) : "v"(x)); return i; }

struct ShadowPos { float px, py, pz; };
// light-space position -> shadow-map coordinates, forward.hlsl:69-74
__device__ __forceinline__ void shadow_coords(float lsx, float lsy, float lsz, float lsw, ShadowPos &p) {
#pragma clang fp contract(off)
    if (__ballot(lsw != 1.0f) == 0ull) { p.px = lsx; p.py = lsy; p.pz = lsz; }   // orthographic sun: w == 1, x / 1 == x
    else { p.px = lsx / lsw; p.py = lsy / lsw; p.pz = lsz / lsw; }
    p.px = p.px * 0.5f + 0.5f;
    p.py = p.py * 0.5f + 0.5f;
    p.py = 1.0f - p.py;
}
// the bounds-table entry that covers all 25 taps of p (byte offset into the table of `pitch` float2 entries per row), or false:
// outside the table's reach
// `margin`: texels that must follow the first texel of tap 0 inside the map.  3 = the window [bx, bx + 3] of full-precision coordinates.
// 4 for the D3D-style sampler (ARCTIC_OPT_SAMPLER bit 2): a coordinate snapped to 1/256 texel may cross an integer, so the window starts at
// bx or bx + 1 and ends at bx + 4 at most -- still inside the entry's texels [4i, 4i + 8) for bx in [4i, 4i + 4).
__device__ __forceinline__ bool shadow_table_offset(uint32_t S, uint32_t pitch, const ShadowPos &p, uint32_t &offset, uint32_t margin = 3u) {
#pragma clang fp contract(off)
    // first texel of tap 0 (u_0 = px - 2e-4) per axis.  0 <= bx < S - 3 means: inside the map with three more texels after it,
    // so 0 < px < 1, no tap wraps, and -- the taps spanning 4e-4 S < 2 texels -- every texel a tap reads lies in [bx, bx + 3]
    const float Sf = (float)S;
    const int bx = floor_to_int((p.px + -0.0002f) * Sf - 0.5f), by = floor_to_int((p.py + -0.0002f) * Sf - 0.5f);
    offset = (((uint32_t)by >> 2) * pitch + ((uint32_t)bx >> 2)) * 8u;
    return (uint32_t)bx < S - margin && (uint32_t)by < S - margin && !(p.pz > 1.0f);
}

}  // namespace arctic
