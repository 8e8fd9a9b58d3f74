// shadow_coords.h -- light-space position -> shadow-map coordinates and the entry of the min/max table that covers a pixel's
// 25 PCF taps: ONE definition for the shading kernels (shade.hip: the test itself) and the G-buffer prepass (geometry.hip: the
// per-tile cost hint, "can a pixel of this tile be lit at all?").  Device code only.
#pragma once
#include "common.h"

namespace arctic {

// floor and convert in one instruction (exact for |x| < 2^31)
__device__ __forceinline__ int floor_to_int(float x) { int i; asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(i) : "v"(x)); return i; }

#ifndef ARCTIC_NEAR_ONE_DIVISION
#define ARCTIC_NEAR_ONE_DIVISION 1   // A/B switch: 0 = three full IEEE divisions whenever a lane's w is not exactly 1 (rounds 1-4)
#endif
struct ShadowPos { float px, py, pz; };
// THE IEEE QUOTIENT x / w FOR A DIVISOR WITHIN SIX ULP-STEPS OF 1.0, in three instructions behind one v_rcp_f32 shared by the three numerators
// (round 5).  The reference's sun is orthographic (scene.cpp:61-70): w is 1 at every vertex and (b0 + b1) + b2 -- one, two, three ulps off 1 --
// after interpolation; in the benchmark frame 30 % of the pixels carry such a w and NOT ONE 8x8 tile is free of them, so every tile paid three
// full IEEE divisions (v_div_scale x 2, v_rcp, 7 x fma, v_div_fmas, v_div_fixup each: 33 VALU + 3 transcendentals, a quarter of a shadowed tile).
// The quotient feeds floor() and the 25 PCF compares: it has to be the IEEE one, bit for bit.  q0 = x y, q = fma(fma(-w, q0, x), y, q0) with
// y = v_rcp_f32(w) IS that for every w = 1.0 -+ 0..6 ulp-steps and EVERY x with 2^-100 < |x| < 2^100 -- all 13 x 2^32 cases compared with the
// compiler's division on the device (tools/experiments/div_near_one.hip, 1 s; tests/test_gpu_division.py runs it) -- and near_one_divisible is that
// domain.  (Markstein's correction step; its textbook exception, a divisor whose mantissa is all ones, is w = 1 - 2^-24 itself: hence enumeration.)
__device__ __forceinline__ float quotient_near_one(float x, float w, float y /* v_rcp_f32(w) */) {
#pragma clang fp contract(off)
    const float q0 = x * y;
    return __builtin_fmaf(__builtin_fmaf(-w, q0, x), y, q0);
}
__device__ __forceinline__ bool near_one_divisible(float x, float y, float z, float w) {
    const float hi = fmaxf(fmaxf(fabsf(x), fabsf(y)), fabsf(z)), lo = fminf(fminf(fabsf(x), fabsf(y)), fabsf(z));
    return __float_as_uint(w) - 0x3F7FFFFAu <= 12u && lo > 0x1p-100f && hi < 0x1p100f;
}
// light-space position -> shadow-map coordinates, forward.hlsl:69-74
__device__ __forceinline__ void shadow_coords(float lsx, float lsy, float lsz, float lsw, ShadowPos &p) {
#pragma clang fp contract(off)
    if (__ballot(lsw != 1.0f) == 0ull) { p.px = lsx; p.py = lsy; p.pz = lsz; }   // w == 1 in every lane, x / 1 == x
    else if (ARCTIC_NEAR_ONE_DIVISION && __ballot(!near_one_divisible(lsx, lsy, lsz, lsw)) == 0ull) {          // orthographic sun: w == 1 -+ a few ulps (nearly every tile)
        const float y = __builtin_amdgcn_rcpf(lsw);
        p.px = quotient_near_one(lsx, lsw, y); p.py = quotient_near_one(lsy, lsw, y); p.pz = quotient_near_one(lsz, lsw, y);
    } else { p.px = lsx / lsw; p.py = lsy / lsw; p.pz = lsz / lsw; }
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
