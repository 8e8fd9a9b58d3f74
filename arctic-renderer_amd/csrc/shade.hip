// shade.hip -- the hot path: per-pixel forward PBR shading + fused tonemap for gfx950.
//
// Replaces ps_main of shaders/forward.hlsl:208-235 (material fetch :98-124, calculate_shadow
// :68-96, Cook-Torrance GGX :126-193, point-light loop :224-231) and main of
// shaders/post_process.hlsl:59-93 (Reinhard / exposure / ACES :39-57, gamma :34-37), plus the skybox lookup of
// shaders/skybox.hlsl:61-90 for pixels without geometry; reads the tile-major G-buffer written by geometry.hip (or, for
// whole frames, the visibility plane: k_material_vis) and stores RGBA8.
//
// Two kernels, each shaped for its own regime (MI355X has no pixel-shader scheduler to do this for us):
//   k_material  1 wavefront = one 8x8 tile (lane l = pixel (l&7, l>>3)), 4 tiles per workgroup.  Reads the tile-major
//               G-buffer with 16 B / lane contiguous loads, fetches the material (software bilinear: address math +
//               global_load + weights; sRGB decode through a 256-entry LDS table; descriptors in LDS) and runs the
//               PCF shadow test.  Memory-latency bound, small register footprint, high occupancy.  Pixels whose
//               (1 - shadow) is 0 are FINISHED here (every light term of ps_main is multiplied by it, point lights
//               included: forward.hlsl:222,230), the others are compacted with a wave-wide ballot into a dense
//               "lit pixel" stream (52 B records).
//   k_light     persistent; 1 lane = one lit pixel of that stream, so no lane idles at shadow boundaries.  Point lights
//               sit in LDS as PAIRS and are evaluated two at a time in packed FP32 (v_pk_fma_f32 ...), the sun first;
//               a pair whose n.wi <= 0 for the whole wave is skipped (exact: the term is multiplied by max(n.wi, 0)).
//               Pure VALU; then tonemap + gamma + RGBA8 store.
// Scalar-per-pixel FP32: no MFMA, by design.
//
// Numerics: texel coordinates/weights and the whole shadow test are computed exactly as the
// CPU oracle does (fp contract off, IEEE divide) because they feed discontinuous decisions; the
// BRDF and tonemap use v_rcp/v_rsq/v_exp/v_log (~1 ulp) and free contraction, inside the 1e-4
// per-channel budget of the output.
#include "common.h"
#include "edges.h"

namespace arctic {

namespace {

struct f3 { float x, y, z; };
__device__ __forceinline__ f3 mk(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 operator*(f3 a, f3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ f3 normalize(f3 v) { return v * rsq(dot(v, v)); }
__device__ __forceinline__ float sat(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, 1.0f); }
__device__ __forceinline__ float pow_fast(float x, float e) {   // x >= 0
    return __builtin_amdgcn_exp2f(e * __builtin_amdgcn_logf(x));
}

// PI = 3.14159265 as written at forward.hlsl:1
constexpr float INV_PI = 1.0f / 3.14159265f;

// ---- sampler: MIN_MAG_MIP_LINEAR + WRAP (forward_pass.cpp:38-51), texel centres at +0.5 ----------
// exact arithmetic: these coordinates select texels and weights
__device__ __forceinline__ void wrap_axis(float u, uint32_t n, int &i0, int &i1, float &f) {
#pragma clang fp contract(off)
    float uw = u - floorf(u);
    float x = uw * (float)n - 0.5f;
    float xf = floorf(x);
    f = x - xf;
    i0 = (int)xf;
    i1 = i0 + 1;
    if (i0 < 0) i0 += (int)n;
    if (i1 >= (int)n) i1 -= (int)n;
}

struct Taps { uint32_t t00, t10, t01, t11; float w00, w10, w01, w11; };

// one texture of a material.  packed = 0: plain RGBA8 image.  packed = 1: the material's three equally sized images
// are stored as ONE image of 8-byte texels holding exactly the eight channels ps_main reads (forward.hlsl:98-124):
//   word 0 = diffuse.r | diffuse.g << 8 | diffuse.b << 16 | normal.r << 24
//   word 1 = normal.g | normal.b << 8 | metal_rough.g << 16 | metal_rough.b << 24
struct TexRef { const uint32_t *texels; uint32_t w, h, packed; };

__device__ __forceinline__ Taps fetch_taps(const TexRef &d, float u, float v) {
    int x0, x1, y0, y1;
    float fx, fy;
    wrap_axis(u, d.w, x0, x1, fx);
    wrap_axis(v, d.h, y0, y1, fy);
    const uint32_t *r0 = d.texels + (size_t)y0 * d.w, *r1 = d.texels + (size_t)y1 * d.w;
    Taps t;
    t.t00 = r0[x0]; t.t10 = r0[x1]; t.t01 = r1[x0]; t.t11 = r1[x1];
    float gx = 1.0f - fx, gy = 1.0f - fy;
    t.w00 = gx * gy; t.w10 = fx * gy; t.w01 = gx * fy; t.w11 = fx * fy;
    return t;
}

// packed material: ONE footprint, four 8-byte loads fetch all three textures' taps; the words are re-cut so that each
// Taps again carries its channels at the byte positions of the plain RGBA8 texel (r = byte 0, g = byte 1, b = byte 2)
__device__ __forceinline__ void fetch_taps3(const TexRef &d, float u, float v, Taps &a, Taps &b, Taps &c) {
    int x0, x1, y0, y1;
    float fx, fy;
    wrap_axis(u, d.w, x0, x1, fx);
    wrap_axis(v, d.h, y0, y1, fy);
    const uint2 *r0 = reinterpret_cast<const uint2 *>(d.texels) + (size_t)y0 * d.w, *r1 = reinterpret_cast<const uint2 *>(d.texels) + (size_t)y1 * d.w;
    const uint2 q00 = r0[x0], q10 = r0[x1], q01 = r1[x0], q11 = r1[x1];
    float gx = 1.0f - fx, gy = 1.0f - fy;
    a.w00 = b.w00 = c.w00 = gx * gy; a.w10 = b.w10 = c.w10 = fx * gy; a.w01 = b.w01 = c.w01 = gx * fy; a.w11 = b.w11 = c.w11 = fx * fy;
    a.t00 = q00.x; a.t10 = q10.x; a.t01 = q01.x; a.t11 = q11.x;                                   // diffuse r,g,b in bytes 0..2
    b.t00 = __builtin_amdgcn_alignbit(q00.y, q00.x, 24); b.t10 = __builtin_amdgcn_alignbit(q10.y, q10.x, 24);   // normal r,g,b
    b.t01 = __builtin_amdgcn_alignbit(q01.y, q01.x, 24); b.t11 = __builtin_amdgcn_alignbit(q11.y, q11.x, 24);
    c.t00 = q00.y >> 8; c.t10 = q10.y >> 8; c.t01 = q01.y >> 8; c.t11 = q11.y >> 8;             // metal-rough g, b in bytes 1, 2
}
__device__ __forceinline__ float ch(uint32_t t, int c) { return (float)((t >> (8 * c)) & 0xFFu); }
__device__ __forceinline__ float filt_unorm(const Taps &t, int c) {
    return (t.w00 * ch(t.t00, c) + t.w10 * ch(t.t10, c) + t.w01 * ch(t.t01, c) + t.w11 * ch(t.t11, c)) * (1.0f / 255.0f);
}
__device__ __forceinline__ float filt_srgb(const Taps &t, int c, const float *lut) {
    return t.w00 * lut[(t.t00 >> (8 * c)) & 0xFFu] + t.w10 * lut[(t.t10 >> (8 * c)) & 0xFFu] +
           t.w01 * lut[(t.t01 >> (8 * c)) & 0xFFu] + t.w11 * lut[(t.t11 >> (8 * c)) & 0xFFu];
}

// ---- forward.hlsl:68-96 calculate_shadow: 5x5 taps, each a bilinear fetch of the R32 map, WRAP -----
// Bit-exact against the oracle: the result is k/25 and one flipped comparison is a visible error.
__device__ __forceinline__ float lerp_exact(float a, float b, float t) { return __builtin_fmaf(t, b - a, a); }

__device__ float shadow_generic(const float *__restrict__ map, uint32_t S, float px, float py, float pz) {
#pragma clang fp contract(off)
    float shadow = 0.0f;
    for (int i = -2; i <= 2; ++i) {
        int x0, x1; float fx;
        wrap_axis(px + (float)i * 0.0001f, S, x0, x1, fx);
        for (int j = -2; j <= 2; ++j) {
            int y0, y1; float fy;
            wrap_axis(py + (float)j * 0.0001f, S, y0, y1, fy);
            const float *r0 = map + (size_t)y0 * S, *r1 = map + (size_t)y1 * S;
            float top = lerp_exact(r0[x0], r0[x1], fx), bot = lerp_exact(r1[x0], r1[x1], fx);
            float closest = lerp_exact(top, bot, fy);
            shadow += pz > closest ? 1.0f : 0.0f;
        }
    }
    return shadow / 25.0f;
}

// 16-byte load from a 4-byte aligned address (gfx950 runs in unaligned-access mode: one global_load_dwordx4)
typedef float float4u __attribute__((ext_vector_type(4), aligned(4)));

__device__ __forceinline__ float sel3(int k, float a, float b, float c) { return k == 0 ? a : (k == 1 ? b : c); }

// Fast PCF.  The 25 taps are 1e-4 apart in uv (0.4 texel at S = 4000), so with their bilinear
// neighbours they touch at most a 4x4 texel window whenever S <= 5000: load it once (4 x 16 B) and
//   (1) every bilinear result lies in [min, max] of the window (fmaf lerps with weights in [0,1)
//       cannot leave the interval of their operands), so pz > max => all 25 taps shadowed and
//       pz <= min => none: decided without filtering, exactly;
//   (2) otherwise evaluate the 25 bilinear compares from registers, in the oracle's operation order
//       (horizontal lerps are shared between taps, which does not change any tap's value).
// Lanes near the map border (WRAP would engage) or with a wider footprint use shadow_generic.
__device__ __forceinline__ float shadow_window(const float *__restrict__ map, uint32_t S, float px, float py, float pz) {
#pragma clang fp contract(off)
    int x0[5], y0[5];
    float fx[5], fy[5];
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        float u = px + (float)(i - 2) * 0.0001f, v = py + (float)(i - 2) * 0.0001f;
        // inside [0,1): u - floor(u) == u, so this is wrap_axis without the wrap
        ok = ok && u >= 0.0f && u < 1.0f && v >= 0.0f && v < 1.0f;
        float x = u * (float)S - 0.5f, y = v * (float)S - 0.5f;
        float xf = floorf(x), yf = floorf(y);
        fx[i] = x - xf; fy[i] = y - yf;
        x0[i] = (int)xf; y0[i] = (int)yf;
    }
    const int bx = x0[0], by = y0[0];
    ok = ok && bx >= 0 && by >= 0 && x0[4] - bx <= 2 && y0[4] - by <= 2 && bx + 3 < (int)S && by + 3 < (int)S;
    if (!ok) return shadow_generic(map, S, px, py, pz);
    const float *base = map + (size_t)by * S + bx;
    const float4u w0 = *reinterpret_cast<const float4u *>(base);
    const float4u w1 = *reinterpret_cast<const float4u *>(base + S);
    const float4u w2 = *reinterpret_cast<const float4u *>(base + 2 * (size_t)S);
    const float4u w3 = *reinterpret_cast<const float4u *>(base + 3 * (size_t)S);
    const float lo = fminf(fminf(fminf(fminf(w0.x, w0.y), fminf(w0.z, w0.w)), fminf(fminf(w1.x, w1.y), fminf(w1.z, w1.w))),
                           fminf(fminf(fminf(w2.x, w2.y), fminf(w2.z, w2.w)), fminf(fminf(w3.x, w3.y), fminf(w3.z, w3.w))));
    const float hi = fmaxf(fmaxf(fmaxf(fmaxf(w0.x, w0.y), fmaxf(w0.z, w0.w)), fmaxf(fmaxf(w1.x, w1.y), fmaxf(w1.z, w1.w))),
                           fmaxf(fmaxf(fmaxf(w2.x, w2.y), fmaxf(w2.z, w2.w)), fmaxf(fmaxf(w3.x, w3.y), fmaxf(w3.z, w3.w))));
    if (pz > hi) return 1.0f;
    if (!(pz > lo)) return 0.0f;
    // horizontal lerps: h[r][i] = lerp(w[r][c_i], w[r][c_i + 1], fx_i)
    float h[4][5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int c = x0[i] - bx;
        h[0][i] = lerp_exact(sel3(c, w0.x, w0.y, w0.z), sel3(c, w0.y, w0.z, w0.w), fx[i]);
        h[1][i] = lerp_exact(sel3(c, w1.x, w1.y, w1.z), sel3(c, w1.y, w1.z, w1.w), fx[i]);
        h[2][i] = lerp_exact(sel3(c, w2.x, w2.y, w2.z), sel3(c, w2.y, w2.z, w2.w), fx[i]);
        h[3][i] = lerp_exact(sel3(c, w3.x, w3.y, w3.z), sel3(c, w3.y, w3.z, w3.w), fx[i]);
    }
    float shadow = 0.0f;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int rr = y0[j] - by;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const float closest = lerp_exact(sel3(rr, h[0][i], h[1][i], h[2][i]), sel3(rr, h[1][i], h[2][i], h[3][i]), fy[j]);
            shadow += pz > closest ? 1.0f : 0.0f;
        }
    }
    return shadow / 25.0f;
}

__device__ __forceinline__ float calculate_shadow(const float *__restrict__ map, uint32_t S, float4 ls) {
#pragma clang fp contract(off)
    if (map == nullptr) return 0.0f;
    float px, py, pz;
    if (__ballot(ls.w != 1.0f) == 0ull) { px = ls.x; py = ls.y; pz = ls.z; }   // orthographic sun: w == 1, x / 1 == x
    else { px = ls.x / ls.w; py = ls.y / ls.w; pz = ls.z / ls.w; }
    px = px * 0.5f + 0.5f;
    py = py * 0.5f + 0.5f;
    py = 1.0f - py;
    if (pz > 1.0f || px < 0.0f || py < 0.0f || px > 1.0f || py > 1.0f) return 0.0f;
    return S <= 5000u ? shadow_window(map, S, px, py, pz) : shadow_generic(map, S, px, py, pz);
}

// ---- forward.hlsl:126-193 -----------------------------------------------------------------------
// calculate_outgoing_radiance with everything that does not depend on the light hoisted into Pix.
// Written twice: scalar (the sun) and for TWO point lights at once on float2 vectors, which hipcc lowers to the packed
// v_pk_{fma,mul,add}_f32 forms.  Measured on MI355X: a wave64 VALU instruction occupies its SIMD for ~4 cycles
// whether packed or not, so the packed loop retires two lights in about the time the scalar one retires one
// (k_light, 64 lights over 8.3 M pixels: 1.09 ms scalar vs 0.7 ms packed).
struct Pix {
    f3 n, wo, world;
    f3 F0, omF0;          // F0 = lerp(0.04, base, metal), 1 - F0                              (:181-182, :128)
    f3 kdb;               // (1 - metal) * base / PI: kD * base / PI = kdb - F * kdb            (:187-192)
    float a2, oma2;       // roughness^4, 1 - roughness^4                                        (:133-139)
    float k, omk;         // k = (roughness + 1)^2 / 8, 1 - k                                    (:147-148)
    float num;            // a2 * g(n.wo) / PI: the light-independent factor of NDF * G          (:131-163)
    float q2, q1, q0;     // (x (1 - k) + k) (4 n.wo x + 1e-4) = q2 x^2 + q1 x + q0, x = n.wi: G's and the BRDF's denominators (:153,:172)
};

__device__ __forceinline__ Pix make_pix(f3 n, f3 wo, f3 world, f3 base, float metal, float rough) {
    Pix p;
    p.n = n; p.wo = wo; p.world = world;
    p.F0 = mk(0.04f + metal * (base.x - 0.04f), 0.04f + metal * (base.y - 0.04f), 0.04f + metal * (base.z - 0.04f));
    p.omF0 = mk(1.0f - p.F0.x, 1.0f - p.F0.y, 1.0f - p.F0.z);
    const float km = (1.0f - metal) * INV_PI;
    p.kdb = mk(base.x * km, base.y * km, base.z * km);
    const float ndwo = fmaxf(dot(n, wo), 0.0f);
    const float a = rough * rough, a2 = a * a;
    p.a2 = a2; p.oma2 = 1.0f - a2;
    const float r1 = rough + 1.0f;
    p.k = r1 * r1 * 0.125f;
    p.omk = 1.0f - p.k;
    p.num = a2 * INV_PI * ndwo * rcp(ndwo * p.omk + p.k);
    const float four_ndwo = 4.0f * ndwo;
    p.q2 = p.omk * four_ndwo; p.q1 = p.k * four_ndwo + 0.0001f * p.omk; p.q0 = 0.0001f * p.k;
    return p;
}

// radiance reflected towards wo from light direction d (unnormalised, towards the light) with colour c, WITHOUT the
// (1 - shadow) factor (common to every light: applied once per pixel).  POINT: radiance = colour / |d|^2
// (forward.hlsl:226-229); otherwise d is unit and there is no falloff.  nd = n . d comes from the culling test.
template <bool POINT>
__device__ __forceinline__ void accumulate_light(const Pix &p, f3 d, float nd, f3 c, f3 &acc) {
    float inv = 1.0f, sc = 1.0f;
    if (POINT) {
        inv = rsq(dot(d, d));
        sc = inv * inv;
    }
    const float ndwi = fmaxf(nd * inv, 0.0f);          // max(n . wi, 0)
    // h = wo + wi formed component-wise like the HLSL (when wi is nearly opposite to wo the sum cancels; the same
    // cancellation keeps the result within rounding distance of the reference arithmetic), left unnormalised
    const f3 h = mk(__builtin_fmaf(d.x, inv, p.wo.x), __builtin_fmaf(d.y, inv, p.wo.y), __builtin_fmaf(d.z, inv, p.wo.z));
    const float hh = dot(h, h), rh = rsq(hh);          // |h|^2, 1 / |h|
    // (h . wo) / |h| = |h| / 2 for unit wo, wi:  clamp(1 - max(h.wo, 0), 0, 1)  (:128, :183)
    const float m = __builtin_fmaf(hh * rh, -0.5f, 1.0f);   // in [0,1] by construction (|h| <= 2): the HLSL clamp only catches rounding
    const float m2 = m * m, p5 = m2 * m2 * m;
    // distribution_ggx's denominator n_dot_h^2 * (a2 - 1) + 1 (:137) cancels to ~a2 at a highlight; written as
    // sin^2 * (1 - a2) + a2 it has no cancellation (same value in exact arithmetic).  sin^2 from e = n - h/|h|:
    // |e|^2 = 2 - 2 cos, sin^2 = |e|^2 (1 - |e|^2 / 4), and n.h > 0 <=> |e|^2 < 2
    const f3 e = mk(__builtin_fmaf(h.x, -rh, p.n.x), __builtin_fmaf(h.y, -rh, p.n.y), __builtin_fmaf(h.z, -rh, p.n.z));   // n - h/|h|
    const float e2 = dot(e, e);
    const float sin2 = e2 * __builtin_fmaf(e2, -0.25f, 1.0f);
    // (when n.h <= 0 the HLSL's max(n.h, 0) makes this 1, but then n.wo <= 0 or n.wi <= 0 and the term is multiplied by 0 anyway)
    const float dd = __builtin_fmaf(sin2, p.oma2, p.a2);
    const float den = (dd * dd) * __builtin_fmaf(__builtin_fmaf(p.q2, ndwi, p.q1), ndwi, p.q0);
    const float spec = (p.num * ndwi) * rcp(den);      // NDF * G / (4 n.wo n.wi + 1e-4)
    sc *= ndwi;
    const f3 F = mk(__builtin_fmaf(p.omF0.x, p5, p.F0.x), __builtin_fmaf(p.omF0.y, p5, p.F0.y), __builtin_fmaf(p.omF0.z, p5, p.F0.z));
    acc.x = __builtin_fmaf(__builtin_fmaf(spec, F.x, __builtin_fmaf(-F.x, p.kdb.x, p.kdb.x)), c.x * sc, acc.x);
    acc.y = __builtin_fmaf(__builtin_fmaf(spec, F.y, __builtin_fmaf(-F.y, p.kdb.y, p.kdb.y)), c.y * sc, acc.y);
    acc.z = __builtin_fmaf(__builtin_fmaf(spec, F.z, __builtin_fmaf(-F.z, p.kdb.z, p.kdb.z)), c.z * sc, acc.z);
}

typedef float v2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2 splat(float x) { v2 r = {x, x}; return r; }
__device__ __forceinline__ v2 fma2(v2 a, v2 b, v2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2 max02(v2 a) { v2 r = {fmaxf(a.x, 0.0f), fmaxf(a.y, 0.0f)}; return r; }
__device__ __forceinline__ v2 sat2(v2 a) { v2 r = {sat(a.x), sat(a.y)}; return r; }
__device__ __forceinline__ v2 rsq2(v2 a) { v2 r = {rsq(a.x), rsq(a.y)}; return r; }
__device__ __forceinline__ v2 rcp2(v2 a) { v2 r = {rcp(a.x), rcp(a.y)}; return r; }

// accumulate_light<true> for two point lights: lane-wise identical arithmetic, each v2 holds {light a, light b}.
// The per-channel tail  (kdb (1 - F) + spec F) colour sc,  F = F0 + omF0 p5,  is linear in three per-light scalars
//     s1 = sc (1 - p5),  s2 = sc spec,  s3 = s2 p5 :   kdb omF0 (colour s1) + F0 (colour s2) + omF0 (colour s3)
// so the loop only accumulates colour x {s1, s2, s3} (9 sums) and the per-pixel factors are applied once after it
// (LightSums::resolve): 12 packed operations per pair instead of 15.
struct LightSums {
    v2 a[3], b[3], c[3];   // sum colour * s1, * s2, * s3 per channel; .x/.y = the two lights of a pair
    __device__ __forceinline__ void clear() { for (int i = 0; i < 3; ++i) { a[i] = splat(0.0f); b[i] = splat(0.0f); c[i] = splat(0.0f); } }
    __device__ __forceinline__ f3 resolve(const Pix &p) const {
        const float A0 = a[0].x + a[0].y, A1 = a[1].x + a[1].y, A2 = a[2].x + a[2].y;
        const float B0 = b[0].x + b[0].y, B1 = b[1].x + b[1].y, B2 = b[2].x + b[2].y;
        const float C0 = c[0].x + c[0].y, C1 = c[1].x + c[1].y, C2 = c[2].x + c[2].y;
        return mk(__builtin_fmaf(p.kdb.x * p.omF0.x, A0, __builtin_fmaf(p.F0.x, B0, p.omF0.x * C0)),
                  __builtin_fmaf(p.kdb.y * p.omF0.y, A1, __builtin_fmaf(p.F0.y, B1, p.omF0.y * C1)),
                  __builtin_fmaf(p.kdb.z * p.omF0.z, A2, __builtin_fmaf(p.F0.z, B2, p.omF0.z * C2)));
    }
};
// d2 = |d|^2 is computed by the caller next to n.d: two independent dependency chains the scheduler can interleave
__device__ __forceinline__ void accumulate_pair(const Pix &p, v2 dx, v2 dy, v2 dz, v2 d2, v2 nd, v2 cr, v2 cg, v2 cb, LightSums &S) {
    const v2 inv = rsq2(d2);
    const v2 ndwi = max02(nd * inv);
    const v2 hx = fma2(dx, inv, splat(p.wo.x)), hy = fma2(dy, inv, splat(p.wo.y)), hz = fma2(dz, inv, splat(p.wo.z));
    const v2 hh = fma2(hz, hz, fma2(hy, hy, hx * hx)), rh = rsq2(hh);
    const v2 m = fma2(hh * rh, splat(-0.5f), splat(1.0f));
    const v2 m2 = m * m, p5 = m2 * m2 * m;
    // sin^2 of the angle between the unit vectors n and h/|h| from their difference e: |e|^2 = 2 - 2 cos, so
    // sin^2 = |e|^2 (1 - |e|^2 / 4) and n.h > 0 <=> |e|^2 < 2.  No cancellation near the highlight (e is small there
    // and carries full relative precision), unlike 1 - (n.h)^2.
    const v2 ex = fma2(hx, -rh, splat(p.n.x)), ey = fma2(hy, -rh, splat(p.n.y)), ez = fma2(hz, -rh, splat(p.n.z));
    const v2 e2 = fma2(ez, ez, fma2(ey, ey, ex * ex));
    // dd = sin^2 (1 - a2) + a2 = a2 + e2 ((1 - a2) - e2 (1 - a2) / 4)
    const v2 dd = fma2(e2, fma2(e2, splat(-0.25f * p.oma2), splat(p.oma2)), splat(p.a2));
    const v2 den = (dd * dd) * fma2(fma2(splat(p.q2), ndwi, splat(p.q1)), ndwi, splat(p.q0));
    const v2 sc = inv * inv * ndwi;
    const v2 s2 = (splat(p.num) * ndwi) * rcp2(den) * sc;   // spec * sc
    const v2 s1 = fma2(-sc, p5, sc), s3 = s2 * p5;
    S.a[0] = fma2(cr, s1, S.a[0]); S.a[1] = fma2(cg, s1, S.a[1]); S.a[2] = fma2(cb, s1, S.a[2]);
    S.b[0] = fma2(cr, s2, S.b[0]); S.b[1] = fma2(cg, s2, S.b[1]); S.b[2] = fma2(cb, s2, S.b[2]);
    S.c[0] = fma2(cr, s3, S.c[0]); S.c[1] = fma2(cg, s3, S.c[1]); S.c[2] = fma2(cb, s3, S.c[2]);
}

// ---- post_process.hlsl ---------------------------------------------------------------------------
__device__ __forceinline__ float rrt_odt(float c) {
    float a = c * (c + 0.0245786f) - 0.000090537f;
    float b = c * (0.983729f * c + 0.4329510f) + 0.238081f;
    return a * rcp(b);
}
__device__ __forceinline__ f3 post_process(f3 c, int tm, float inv_gamma, float exposure) {
    f3 t;
    if (tm == 1) {          // tm_exposure :44-47: 1 - exp(-c * exposure); for tiny arguments the subtraction cancels in
                            // fp32 (1 - exp(-1e-8) = 0), so the series x - x^2/2 + x^3/6 takes over below 1/64
        const float LOG2E = 1.4426950408889634f;
        const f3 x = mk(c.x * exposure, c.y * exposure, c.z * exposure);
        auto one_minus_exp = [&](float v) {
            const float direct = 1.0f - __builtin_amdgcn_exp2f(-v * LOG2E);
            const float series = v * (1.0f + v * (-0.5f + v * (1.0f / 6.0f)));
            return fabsf(v) < 0.015625f ? series : direct;
        };
        t = mk(one_minus_exp(x.x), one_minus_exp(x.y), one_minus_exp(x.z));
    } else if (tm == 2) {   // tm_aces :15-25, :50-57
        f3 i = mk(0.59719f * c.x + 0.35458f * c.y + 0.04823f * c.z, 0.07600f * c.x + 0.90834f * c.y + 0.01566f * c.z,
                  0.02840f * c.x + 0.13383f * c.y + 0.837f * c.z);
        i = mk(rrt_odt(i.x), rrt_odt(i.y), rrt_odt(i.z));
        t = mk(sat(1.60475f * i.x - 0.53108f * i.y - 0.07367f * i.z), sat(-0.10208f * i.x + 1.10813f * i.y - 0.00605f * i.z),
               sat(-0.00327f * i.x - 0.07276f * i.y + 1.07f * i.z));
    } else {                // tm_reinhard :39-42 (and `default:`)
        t = mk(c.x * rcp(c.x + 1.0f), c.y * rcp(c.y + 1.0f), c.z * rcp(c.z + 1.0f));
    }
    // correct_gamma :34-37: pow(abs(c), 1/gamma)
    return mk(pow_fast(fabsf(t.x), inv_gamma), pow_fast(fabsf(t.y), inv_gamma), pow_fast(fabsf(t.z), inv_gamma));
}
// store to the R8G8B8A8_UNORM target (renderer.cpp:161-175): saturate (NaN -> 0), *255, +0.5, truncate
__device__ __forceinline__ uint32_t unorm8(float x) {
    x = x > 0.0f ? x : 0.0f;
    x = x > 1.0f ? 1.0f : x;
    return (uint32_t)(x * 255.0f + 0.5f);
}

// first wave of G-buffer loads: what every pixel needs (28 B)
struct TileHead { float4 a; float b0, b1, b2; };   // a = uv.xy, ls.xy; b = ls.z, ls.w, material id
__device__ __forceinline__ TileHead load_head(const GBuffer &g, size_t idx) {
    TileHead t;
    t.a = g.a[idx];
    t.b0 = g.b[idx * 3]; t.b1 = g.b[idx * 3 + 1]; t.b2 = g.b[idx * 3 + 2];
    return t;
}

__device__ __forceinline__ TexRef lds_desc(const uint4 *d) {
    const uint4 v = *d;
    TexRef t;
    t.texels = reinterpret_cast<const uint32_t *>(((unsigned long long)v.y << 32) | v.x);
    t.w = v.z & 0x7FFFFFFFu; t.h = v.w;
    t.packed = v.z >> 31;   // TexDesc::w bit 31
    return t;
}

// the reference renders ps_main into an R16G16B16A16_FLOAT target (forward_pass.cpp:149) that post_process then reads:
// hdr16 reproduces that rounding (round-to-nearest-even to binary16, finite overflow to +inf like the ROP's conversion)
__device__ __forceinline__ float through_half(float x) { return (float)(_Float16)x; }

__device__ __forceinline__ void store_pixel(const ShadeParams &sp, size_t o, f3 color) {
    if (sp.hdr16) color = mk(through_half(color.x), through_half(color.y), through_half(color.z));
    const f3 l = (sp.debug & 4) ? color : post_process(color, sp.tm_method, sp.inv_gamma, sp.exposure);   // bit 2: timing only
    reinterpret_cast<uint32_t *>(sp.out_rgba8)[o] = unorm8(l.x) | (unorm8(l.y) << 8) | (unorm8(l.z) << 16) | 0xFF000000u;
    if (sp.out_ldr) { sp.out_ldr[o * 3] = l.x; sp.out_ldr[o * 3 + 1] = l.y; sp.out_ldr[o * 3 + 2] = l.z; }
    if (sp.out_hdr) { sp.out_hdr[o * 3] = color.x; sp.out_hdr[o * 3 + 1] = color.y; sp.out_hdr[o * 3 + 2] = color.z; }
}

// ---- skybox.hlsl:61-90: pixels without geometry take the environment map along their view ray --------------------------
// (the reference draws a cube at z = w after the forward pass, depth LESS_EQUAL: it survives exactly where nothing was drawn.)
// The lookup coordinates are computed in float64: they feed a bilinear filter over an HDR image whose texel-to-texel contrast
// can be thousands, so fp32 atan2/asin noise (1e-7 in uv = 2e-4 texel) would show; sky pixels are few, fp64 is affordable.
__device__ __forceinline__ void wrap_axis64(double u, uint32_t n, int &i0, int &i1, float &f) {
    const double uw = u - floor(u);
    const double x = uw * (double)n - 0.5;
    const double xf = floor(x);
    f = (float)(x - xf);
    i0 = (int)xf;
    i1 = i0 + 1;
    if (i0 < 0) i0 += (int)n;
    if (i1 >= (int)n) i1 -= (int)n;
}
__device__ __noinline__ f3 sample_environment(const float4 *__restrict__ env, uint32_t w, uint32_t h, float fx_, float fy_, float fz_) {
    double x = fx_, y = fy_, z = fz_;
    const double inv = 1.0 / sqrt(x * x + y * y + z * z);            // dir = normalize(dir)
    x *= inv; y *= inv; z *= inv;
    const double u = atan2(z, x) * (double)0.1591f + 0.5;             // uv = (atan2(z,x), asin(y)) * INV_ATAN + 0.5
    const double v = -(asin(fmin(fmax(y, -1.0), 1.0)) * (double)0.3183f + 0.5);   // uv.y = -uv.y
    int x0, x1, y0, y1;
    float fx, fy;
    wrap_axis64(u, w, x0, x1, fx);
    wrap_axis64(v, h, y0, y1, fy);
    const float4 a = env[(size_t)y0 * w + x0], b = env[(size_t)y0 * w + x1], c = env[(size_t)y1 * w + x0], d = env[(size_t)y1 * w + x1];
    const float gx = 1.0f - fx, gy = 1.0f - fy;
    const float w00 = gx * gy, w10 = fx * gy, w01 = gx * fy, w11 = fx * fy;
    return mk(w00 * a.x + w10 * b.x + w01 * c.x + w11 * d.x, w00 * a.y + w10 * b.y + w01 * c.y + w11 * d.y,
              w00 * a.z + w10 * b.z + w01 * c.z + w11 * d.z);
}

// ---- the material half of ps_main for one 8x8 tile (one wave) ------------------------------------------------------
// forward.hlsl:98-124 (material fetch) + :64-96 (shadow test) + classification.  Pixels that need no light loop are
// finished here (no geometry: skybox or black; fully shadowed: ambient * base); for every other pixel (`live`) the lane
// gets the record the light loop needs.  Returns the ballot of live lanes.  Shared by k_material (two-kernel pass) and
// k_shade_fused.
struct LitRec { float4 r0, r1, r2; uint32_t px; };   // world.xyz, 1 - shadow | n.xyz, roughness | base.rgb, metalness | output index
// `second(gc, gd, ge)` delivers the lit pixels' remaining attributes (world position + tangent frame, packed like the
// G-buffer planes c, d, e): loaded from the G-buffer, or interpolated on the spot by the visibility-buffer kernel.
template <class Second>
__device__ __forceinline__ unsigned long long material_tile(const ShadeParams &sp, const float *lut, const uint4 *ldesc, uint32_t ty, uint32_t tx,
                                                            uint32_t lane, const TileHead &cur, bool &live, LitRec &rec, Second second) {
    const uint32_t x = tx * 8 + (lane & 7);
    const int32_t y = (int32_t)(ty * 8 + (lane >> 3)) - (int32_t)sp.row0_in_tile;
    const bool in_frame = x < sp.width && y >= 0 && y < (int32_t)sp.rows;
    const uint32_t mat = __float_as_uint(cur.b2);
    const bool covered = in_frame && mat < sp.n_materials;
    const size_t o = (size_t)y * sp.width + x;

    // ---- material fetch, forward.hlsl:98-124, and the shadow test: needs only uv, light-space position, material --------
    Taps t0, t1, t2;
    float lit = 0.0f;
    if (covered) {
        const float u = cur.a.x, v = cur.a.y;
        if (!(sp.debug & 2)) lit = 1.0f - calculate_shadow(sp.shadow_map, sp.shadow_size, make_float4(cur.a.z, cur.a.w, cur.b0, cur.b1));
        const TexRef d0 = lds_desc(ldesc + mat * 3);
        if (sp.debug & 1) { t0.t00 = t0.t10 = t0.t01 = t0.t11 = 0x808080u; t0.w00 = t0.w10 = t0.w01 = t0.w11 = 0.25f; t1 = t0; t2 = t0; }   // timing only
        else if (d0.packed) fetch_taps3(d0, u, v, t0, t1, t2);   // the usual case (equal-size images); waves mixing both kinds diverge
        else { t0 = fetch_taps(d0, u, v); t1 = fetch_taps(lds_desc(ldesc + mat * 3 + 1), u, v); t2 = fetch_taps(lds_desc(ldesc + mat * 3 + 2), u, v); }
    }
    // exact culling 1: Lo of ps_main is a sum of terms each multiplied by (1 - shadow) (point lights too: forward.hlsl:222,
    // 230), so a fully shadowed pixel is ambient * base and needs neither the sun, nor any point light, nor its normal,
    // tangent frame, position, metalness or roughness.  Everything else goes to the light loop.
    live = covered && (sp.culling ? lit != 0.0f : true);
    const unsigned long long m = __ballot(live);
    float4 gc, gd, ge;
    if (live) second(gc, gd, ge);   // second wave: lit pixels only (48 B / pixel from the G-buffer, whole 128-byte tile rows)
    f3 base = mk(0.0f, 0.0f, 0.0f);
    if (covered) base = mk(filt_srgb(t0, 0, lut), filt_srgb(t0, 1, lut), filt_srgb(t0, 2, lut));
    if (in_frame && !live) {
        f3 c = base * sp.ambient;   // covered and fully shadowed: ambient * base
        if (!covered && sp.env) {   // no geometry: the skybox
            const int gy = (row_global((int)ty, sp.band_tiles, sp.shard_count, sp.shard_index) + sp.tile_y0) * 8 + (int)(lane >> 3);
            const float nx = __builtin_fmaf((float)x + 0.5f, sp.ndc_sx, -1.0f), ny = __builtin_fmaf(-((float)gy + 0.5f), sp.ndc_sy, 1.0f);
            c = sample_environment(sp.env, sp.env_w, sp.env_h,
                                   __builtin_fmaf(sp.sky_up[0], ny, __builtin_fmaf(sp.sky_right[0], nx, sp.sky_fwd[0])),
                                   __builtin_fmaf(sp.sky_up[1], ny, __builtin_fmaf(sp.sky_right[1], nx, sp.sky_fwd[1])),
                                   __builtin_fmaf(sp.sky_up[2], ny, __builtin_fmaf(sp.sky_right[2], nx, sp.sky_fwd[2])));
        }
        store_pixel(sp, o, c);
    }
    if (live) {
        float r = filt_unorm(t1, 0), g = 1.0f - filt_unorm(t1, 1), b = filt_unorm(t1, 2);   // normal.g = 1 - normal.g
        r = r * 2.0f - 1.0f; g = g * 2.0f - 1.0f; b = b * 2.0f - 1.0f;
        // mul(tbn, v), tbn columns t, b, n
        const f3 T = mk(gc.w, gd.x, gd.y), B = mk(gd.z, gd.w, ge.x), N = mk(ge.y, ge.z, ge.w);
        const f3 n = normalize(T * r + B * g + N * b);
        const float rough = filt_unorm(t2, 1), metal = filt_unorm(t2, 2);   // .g, .b (forward.hlsl:117,123)
        rec.r0 = make_float4(gc.x, gc.y, gc.z, lit);
        rec.r1 = make_float4(n.x, n.y, n.z, rough);
        rec.r2 = make_float4(base.x, base.y, base.z, metal);
        rec.px = (uint32_t)o;
    }
    return m;
}

__device__ __forceinline__ void stage_material_lds(const ShadeParams &sp, float *lut, uint4 *ldesc) {
    lut[threadIdx.x] = sp.srgb_lut[threadIdx.x];
    for (uint32_t i = threadIdx.x; i < sp.n_materials * 3; i += 256) ldesc[i] = reinterpret_cast<const uint4 *>(sp.tex)[i];
}

// what happens to the live pixels of a tile -- three ways, chosen per pass by the host (ShadeLaunch::inline_mode):
//   MODE 2  the light loop runs right here, packed pairs from LDS (light_pixel<2>).  Lit regions are contiguous, so most
//           lit tiles are fully lit and the lanes idling at shadow boundaries cost less than the alternative's 52-byte
//           record written and read back per lit pixel plus a second kernel: the default above 16 point lights
//           (4K, 64 lights: 0.259 ms against 0.273 ms; still ahead at 256 lights).
//   MODE 1  the same with a scalar loop over the lights read through the scalar cache: 66 VGPRs keep the occupancy of the
//           memory-bound part (7 waves/SIMD against 5), which is what matters with few lights: the default up to 16
//           (the reference's MAX_NUM_POINT_LIGHTS); sun only: 0.121 ms = 68 % of the HBM roof.
//   MODE 0  the live pixels are compacted into the lit-pixel stream for k_light (one atomicAdd per wave, on one of
//           LIT_SHARDS counters each on its own 128-byte line: a single counter would serialise at ~88 atomics/us):
//           every lane of k_light is busy; used for the light-evaluation statistics and kept as an option.
// All three evaluate the same formulas; images agree to fp32 rounding (the compiler contracts differently per kernel).
template <int LIGHTS_PER_TRIP>
__device__ __forceinline__ void light_pixel(const ShadeParams &sp, const float4 *llights, uint32_t lane, float4 r0, float4 r1, float4 r2, uint32_t o);
__device__ __forceinline__ void stage_lights_lds(const ShadeParams &sp, float4 *llights);

template <int MODE>
__device__ __forceinline__ void emit_live(const ShadeParams &sp, uint32_t ty, uint32_t tx, uint32_t bpr, uint32_t lane, unsigned long long m,
                                          bool live, const LitRec &rec, const float4 *llights = nullptr) {
    if (m == 0ull) return;
    if (MODE == 2) {
        if (live) light_pixel<2>(sp, llights, lane, rec.r0, rec.r1, rec.r2, rec.px);
        return;
    }
    if (MODE == 1) {
        if (live) {
            const f3 world = mk(rec.r0.x, rec.r0.y, rec.r0.z), n = mk(rec.r1.x, rec.r1.y, rec.r1.z), base = mk(rec.r2.x, rec.r2.y, rec.r2.z);
            const f3 wo = normalize(mk(sp.eye[0], sp.eye[1], sp.eye[2]) - world);
            const Pix px = make_pix(n, wo, world, base, rec.r2.w, rec.r1.w);
            f3 sun = mk(0.0f, 0.0f, 0.0f);
            const f3 d = mk(-sp.sun_dir[0], -sp.sun_dir[1], -sp.sun_dir[2]);
            accumulate_light<false>(px, d, dot(n, d), mk(sp.sun_color[0], sp.sun_color[1], sp.sun_color[2]), sun);
            for (uint32_t i = 0; i < sp.n_lights; ++i) {   // a handful of point lights
                const float4 lp = sp.lights[2 * i], lc = sp.lights[2 * i + 1];   // wave-uniform: scalar loads
                const f3 dl = mk(lp.x, lp.y, lp.z) - world;
                const float ndl = dot(n, dl);
                if (sp.culling && ndl <= 0.0f) continue;      // n.wi <= 0: the term is multiplied by max(n.wi, 0) = 0
                accumulate_light<true>(px, dl, ndl, mk(lc.x, lc.y, lc.z), sun);
            }
            store_pixel(sp, rec.px, sun * rec.r0.w + base * sp.ambient);   // r0.w = 1 - shadow
        }
        return;
    }
    const uint32_t shard = (ty * bpr + (tx >> 2)) % LIT_SHARDS;   // by screen position: lit regions spread over all shards
    uint32_t first = (uint32_t)__ffsll((long long)m) - 1, slot0 = 0;
    if (lane == first) slot0 = atomicAdd(sp.lit_count + (sp.band * LIT_SHARDS + shard) * LIT_COUNTER_STRIDE, (uint32_t)__popcll(m));
    slot0 = __shfl(slot0, (int)first);
    if (live) {
        const size_t slot = (size_t)(sp.band * LIT_SHARDS + shard) * sp.lit_shard_cap + slot0 + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        sp.lit_r0[slot] = rec.r0;
        sp.lit_r1[slot] = rec.r1;
        sp.lit_r2[slot] = rec.r2;
        sp.lit_px[slot] = rec.px;
    }
}

// ---- kernel 1 of the two-kernel pass: material_tile over every tile, live pixels appended to the lit-pixel stream -----
// LDS (dynamic): [0,256) sRGB LUT | texture descriptors, 4 dwords each
template <int MODE>
__global__ __launch_bounds__(256) void k_material(const ShadeParams sp) {
    extern __shared__ __align__(16) float smem[];
    float *lut = smem;
    uint4 *ldesc = reinterpret_cast<uint4 *>(smem + 256);
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // XCD-aware order: workgroups are dealt round-robin over the 8 XCDs (each with its own 4 MiB L2), so physical block b
    // runs on XCD b % 8.  A workgroup = 4 horizontally adjacent tiles; XCD x takes the tile rows y = x (mod 8), walking
    // each row left to right, so horizontal neighbours -- which share texture and shadow-map cache lines -- meet in the
    // same L2, while all eight XCDs stay within 8 tile rows of each other in the G-buffer stream.  (Placement is a speed
    // matter only; the grid is padded to whole groups of 8 rows and surplus blocks exit.)
    const uint32_t bpr = (sp.tiles_x + 3) >> 2;                       // workgroups per tile row
    const uint32_t xcd = blockIdx.x & 7u, idx = blockIdx.x >> 3;
    const uint32_t ty = ((idx / bpr) * sp.n_bands + sp.band) * 8 + xcd, tx = (idx % bpr) * 4 + wave;   // band: see launch_shade
    const bool tile_ok = ty < sp.tiles_y && tx < sp.tiles_x;
    TileHead cur;
    if (tile_ok) cur = load_head(sp.g, ((size_t)ty * sp.tiles_x + tx) * 64 + lane);   // in flight while LDS is staged
    stage_material_lds(sp, lut, ldesc);
    float4 *llights = reinterpret_cast<float4 *>(smem + 256 + (size_t)sp.n_materials * 12);
    if (MODE == 2) stage_lights_lds(sp, llights);
    __syncthreads();
    if (!tile_ok) return;
    bool live;
    LitRec rec;
    const size_t gi = ((size_t)ty * sp.tiles_x + tx) * 64 + lane;
    const unsigned long long m = material_tile(sp, lut, ldesc, ty, tx, lane, cur, live, rec,
                                               [&](float4 &gc, float4 &gd, float4 &ge) { gc = sp.g.c[gi]; gd = sp.g.d[gi]; ge = sp.g.e[gi]; });
    emit_live<MODE>(sp, ty, tx, bpr, lane, m, live, rec, llights);
}

// ---- kernel 1 without a G-buffer (whole frames): the same tile walk straight from the visibility plane ----------------
// arctic_render_frame has no use for the 76 B/pixel G-buffer between its own two kernels: writing it (k_resolve, 630 MB at
// 4K) and reading it back costs more than interpolating again.  This variant reads the 8-byte visibility key, finds the
// triangle, and interpolates uv + light-space position for every covered pixel and world position + tangent frame only
// for the lit ones -- with the very operations of k_resolve (edges.h, fp contraction off), so the pixels are bit-identical
// to the G-buffer path.  Everything after the attributes is material_tile, shared.
template <int MODE>
__global__ __launch_bounds__(256) void k_material_vis(const ShadeParams sp) {
    extern __shared__ __align__(16) float smem[];
    float *lut = smem;
    uint4 *ldesc = reinterpret_cast<uint4 *>(smem + 256);
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t bpr = (sp.tiles_x + 3) >> 2;
    const uint32_t xcd = blockIdx.x & 7u, idx = blockIdx.x >> 3;   // XCD-aware order: see k_material
    const uint32_t ty = ((idx / bpr) * sp.n_bands + sp.band) * 8 + xcd, tx = (idx % bpr) * 4 + wave;
    const bool tile_ok = ty < sp.tiles_y && tx < sp.tiles_x;
    const size_t gi = ((size_t)ty * sp.tiles_x + tx) * 64 + lane;
    unsigned long long key = ~0ull;
    if (tile_ok) key = sp.vis[gi];
    stage_material_lds(sp, lut, ldesc);
    float4 *llights = reinterpret_cast<float4 *>(smem + 256 + (size_t)sp.n_materials * 12);
    if (MODE == 2) stage_lights_lds(sp, llights);
    __syncthreads();
    if (!tile_ok) return;
    const int32_t px = (int32_t)(tx * 8 + (lane & 7));
    const int32_t py = (row_global((int)ty, sp.band_tiles, sp.shard_count, sp.shard_index) + sp.tile_y0) * 8 + (int32_t)(lane >> 3);
    TileHead cur;
    cur.a = make_float4(0.0f, 0.0f, 0.0f, 0.0f); cur.b0 = 0.0f; cur.b1 = 0.0f; cur.b2 = __uint_as_float(NO_MATERIAL);
    float B[3] = {0.0f, 0.0f, 0.0f};
    const float *A0 = nullptr, *A1 = nullptr, *A2 = nullptr;
    if (key != ~0ull) {
        const SetupRec &t = sp.recs[sp.rec_of[(uint32_t)key]];   // low word of the key = order id (k_setup)
        source_barycentrics(t, px, py, B);
        const ObjectRec &ob = sp.objs[t.object];
        const uint32_t lt = t.src_tri - ob.first_triangle;
        A0 = sp.xv[ob.first_xvert + ob.indices[3 * lt]].attr;
        A1 = sp.xv[ob.first_xvert + ob.indices[3 * lt + 1]].attr;
        A2 = sp.xv[ob.first_xvert + ob.indices[3 * lt + 2]].attr;
        cur.a = make_float4(interpolate_attr(B, A0, A1, A2, 0), interpolate_attr(B, A0, A1, A2, 1),
                            interpolate_attr(B, A0, A1, A2, 14), interpolate_attr(B, A0, A1, A2, 15));
        cur.b0 = interpolate_attr(B, A0, A1, A2, 16); cur.b1 = interpolate_attr(B, A0, A1, A2, 17);
        cur.b2 = __uint_as_float(ob.material);
    }
    bool live;
    LitRec rec;
    const unsigned long long m = material_tile(sp, lut, ldesc, ty, tx, lane, cur, live, rec, [&](float4 &gc, float4 &gd, float4 &ge) {
        // attribute order (XVert::attr): uv 0-1, t 2-4, b 5-7, n 8-10, world 11-13, light space 14-17; planes as gbuffer_pack
        gc = make_float4(interpolate_attr(B, A0, A1, A2, 11), interpolate_attr(B, A0, A1, A2, 12), interpolate_attr(B, A0, A1, A2, 13), interpolate_attr(B, A0, A1, A2, 2));
        gd = make_float4(interpolate_attr(B, A0, A1, A2, 3), interpolate_attr(B, A0, A1, A2, 4), interpolate_attr(B, A0, A1, A2, 5), interpolate_attr(B, A0, A1, A2, 6));
        ge = make_float4(interpolate_attr(B, A0, A1, A2, 7), interpolate_attr(B, A0, A1, A2, 8), interpolate_attr(B, A0, A1, A2, 9), interpolate_attr(B, A0, A1, A2, 10));
    });
    emit_live<MODE>(sp, ty, tx, bpr, lane, m, live, rec, llights);
}

// ---- the light half of ps_main for one lit pixel per lane: the sun + every point light, tonemap, store --------------
// LDS image of the point lights: PAIRS, 12 floats per pair {x0,x1, y0,y1, z0,z1, r0,r1, g0,g1, b0,b1}; every lane reads
// the same address, so a pair costs three broadcast ds_read_b128.  The count is padded to a multiple of 4 with black lights.
__device__ __forceinline__ void stage_lights_lds(const ShadeParams &sp, float4 *llights) {
    const uint32_t n_pairs = 2 * ((sp.n_lights + 3) >> 2);
    for (uint32_t i = threadIdx.x; i < 2 * n_pairs; i += 256) {
        float4 lp = make_float4(0.0f, 1.0e6f, 0.0f, 0.0f), lc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);   // pad: colour 0
        if (i < sp.n_lights) { lp = sp.lights[2 * i]; lc = sp.lights[2 * i + 1]; }
        float *dst = reinterpret_cast<float *>(llights) + (size_t)(i >> 1) * 12 + (i & 1);
        dst[0] = lp.x; dst[2] = lp.y; dst[4] = lp.z; dst[6] = lc.x; dst[8] = lc.y; dst[10] = lc.z;
    }
}

template <int LIGHTS_PER_TRIP>
__device__ __forceinline__ void light_pixel(const ShadeParams &sp, const float4 *llights, uint32_t lane, float4 r0, float4 r1, float4 r2, uint32_t o) {
    const uint32_t n_quads = (sp.n_lights + 3) >> 2, n_pairs = 2 * n_quads;
    const f3 eye = mk(sp.eye[0], sp.eye[1], sp.eye[2]);
    const f3 world = mk(r0.x, r0.y, r0.z), n = mk(r1.x, r1.y, r1.z), base = mk(r2.x, r2.y, r2.z);
    const f3 wo = normalize(eye - world);
    const Pix px = make_pix(n, wo, world, base, r2.w, r1.w);
    f3 sun = mk(0.0f, 0.0f, 0.0f);
    {   // the sun: wi = -sun_dir, radiance = sun_color (forward.hlsl:221-222)
        const f3 d = mk(-sp.sun_dir[0], -sp.sun_dir[1], -sp.sun_dir[2]);
        accumulate_light<false>(px, d, dot(n, d), mk(sp.sun_color[0], sp.sun_color[1], sp.sun_color[2]), sun);
    }
    LightSums S;
    S.clear();
    const v2 wx = splat(world.x), wy = splat(world.y), wz = splat(world.z);
    // two pairs (four lights) per trip: the two evaluations are independent, which gives the scheduler instructions to
    // put between dependent packed operations (n_quads = ceil(n_pairs / 2); the LDS image is padded with black lights)
    if (LIGHTS_PER_TRIP == 4)
    for (uint32_t q = 0; q < n_quads; ++q) {
        const float4 A0 = llights[6 * q], B0 = llights[6 * q + 1], C0 = llights[6 * q + 2];
        const float4 A1 = llights[6 * q + 3], B1 = llights[6 * q + 4], C1 = llights[6 * q + 5];
        const v2 dx0 = (v2){A0.x, A0.y} - wx, dy0 = (v2){A0.z, A0.w} - wy, dz0 = (v2){B0.x, B0.y} - wz;
        const v2 dx1 = (v2){A1.x, A1.y} - wx, dy1 = (v2){A1.z, A1.w} - wy, dz1 = (v2){B1.x, B1.y} - wz;
        const v2 nd0 = fma2(splat(n.z), dz0, fma2(splat(n.y), dy0, splat(n.x) * dx0)), d20 = fma2(dz0, dz0, fma2(dy0, dy0, dx0 * dx0));
        const v2 nd1 = fma2(splat(n.z), dz1, fma2(splat(n.y), dy1, splat(n.x) * dx1)), d21 = fma2(dz1, dz1, fma2(dy1, dy1, dx1 * dx1));
        // exact culling 2: n.wi <= 0 zeroes a light (forward.hlsl:191-192); skip the trip when that holds for all four
        // lights in every lane of the wave
        if (sp.culling && __ballot(nd0.x > 0.0f || nd0.y > 0.0f || nd1.x > 0.0f || nd1.y > 0.0f) == 0ull) continue;
        accumulate_pair(px, dx0, dy0, dz0, d20, nd0, (v2){B0.z, B0.w}, (v2){C0.x, C0.y}, (v2){C0.z, C0.w}, S);
        accumulate_pair(px, dx1, dy1, dz1, d21, nd1, (v2){B1.z, B1.w}, (v2){C1.x, C1.y}, (v2){C1.z, C1.w}, S);
        if (sp.light_evals) {
            const unsigned long long active = __ballot(1);
            const uint32_t k = min(4u, sp.n_lights - 4 * q);
            if (lane == (uint32_t)__ffsll((long long)active) - 1) atomicAdd(sp.light_evals, (unsigned long long)__popcll(active) * k);
        }
    }
    if (LIGHTS_PER_TRIP == 2)
    for (uint32_t p = 0; p < n_pairs; ++p) {
        const float4 A = llights[3 * p], Bq = llights[3 * p + 1], C = llights[3 * p + 2];
        const v2 dx = (v2){A.x, A.y} - wx, dy = (v2){A.z, A.w} - wy, dz = (v2){Bq.x, Bq.y} - wz;
        // n.d and |d|^2 interleaved by hand: a v_pk_fma that consumes the previous one's result costs a wait state
        const v2 t0 = splat(n.x) * dx, u0 = dx * dx;
        const v2 t1 = fma2(splat(n.y), dy, t0), u1 = fma2(dy, dy, u0);
        const v2 nd = fma2(splat(n.z), dz, t1);
        v2 d2 = fma2(dz, dz, u1);
        asm volatile("" : "+v"(d2));   // keep |d|^2 up here (the compiler would sink it below the branch, back into one serial chain)
        if (sp.culling && __ballot(nd.x > 0.0f || nd.y > 0.0f) == 0ull) continue;
        accumulate_pair(px, dx, dy, dz, d2, nd, (v2){Bq.z, Bq.w}, (v2){C.x, C.y}, (v2){C.z, C.w}, S);
        if (sp.light_evals) {
            const unsigned long long active = __ballot(1);
            const uint32_t k = min(2u, sp.n_lights > 2 * p ? sp.n_lights - 2 * p : 0u);
            if (lane == (uint32_t)__ffsll((long long)active) - 1) atomicAdd(sp.light_evals, (unsigned long long)__popcll(active) * k);
        }
    }
    store_pixel(sp, o, (sun + S.resolve(px)) * r0.w + base * sp.ambient);   // r0.w = 1 - shadow
}

// ---- kernel 2 of the two-kernel pass: light_pixel over the lit-pixel stream ------------------------------------------
// LDS (dynamic): the light pairs.  Persistent: gridDim.x workgroups; wave w works on shard w % LIT_SHARDS of the stream
// and takes every (n_waves / LIT_SHARDS)-th 64-pixel group of it.
template <int LIGHTS_PER_TRIP>
__global__ __launch_bounds__(256) void k_light(const ShadeParams sp) {
    extern __shared__ __align__(16) float smem[];
    float4 *llights = reinterpret_cast<float4 *>(smem);
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    stage_lights_lds(sp, llights);
    __syncthreads();
    // the stream counters are double-buffered: this pass reads the set k_material has just filled and clears the other
    // one for the next pass's k_material -- no memset launch between passes
    if (blockIdx.x == 0)
        for (uint32_t i = threadIdx.x; i < sp.n_bands * LIT_SHARDS; i += 256) sp.lit_count_next[i * LIT_COUNTER_STRIDE] = 0;
    const uint32_t w = blockIdx.x * 4 + wave, n_waves = gridDim.x * 4;
    const bool wide = n_waves >= LIT_SHARDS;
    const uint32_t nsub = wide ? n_waves / LIT_SHARDS : 1u, sub = wide ? w / LIT_SHARDS : 0u;
    if (sub >= nsub) return;
    for (uint32_t shard = wide ? w % LIT_SHARDS : w; shard < LIT_SHARDS; shard += wide ? LIT_SHARDS : n_waves)
    for (uint32_t count = sp.lit_count[(sp.band * LIT_SHARDS + shard) * LIT_COUNTER_STRIDE], grp = sub; grp * 64 < count; grp += nsub) {
        if (grp * 64 + lane >= count) continue;
        const size_t i = (size_t)(sp.band * LIT_SHARDS + shard) * sp.lit_shard_cap + grp * 64 + lane;
        light_pixel<LIGHTS_PER_TRIP>(sp, llights, lane, sp.lit_r0[i], sp.lit_r1[i], sp.lit_r2[i], sp.lit_px[i]);
    }
}

// ---- the whole pass in ONE persistent kernel: material_tile and light_pixel decoupled through a per-wave LDS queue ----
// k_material is bound by memory and k_light by the FP32 VALU; run back to back each leaves the other resource idle, and
// the lit-pixel stream between them costs 52 B written + 52 B read per lit pixel.  Here every wave takes tiles from a
// ticket counter, runs the material half, and appends the live pixels to ITS OWN queue in LDS (128 records); whenever
// the queue holds 64 it pops them and runs the light half with all 64 lanes busy.  Waves of one SIMD are in different
// phases at any moment, so memory waits of one overlap the arithmetic of another, and the stream never touches HBM.
// No wave ever waits for another one (no barrier after the prologue, no inter-wave flag): nothing can deadlock.
// Tickets: one counter per XCD (128 B apart); XCD x walks the tile rows y = x (mod 8) left to right like k_material,
// then steals from the other XCDs' rows.  The last wave to leave resets the counters for the next pass.
// LDS (dynamic): sRGB LUT | texture descriptors | light pairs | 4 queues x 128 x (3 float4 + 1 dword)
constexpr uint32_t FQ_CAP = 128, TICKET_STRIDE = 32, FT_BATCH = 4;
template <int LIGHTS_PER_TRIP>
__global__ __launch_bounds__(256) void k_shade_fused(const ShadeParams sp) {
    extern __shared__ __align__(16) float smem[];
    float *lut = smem;
    uint4 *ldesc = reinterpret_cast<uint4 *>(smem + 256);
    float4 *llights = reinterpret_cast<float4 *>(smem + 256 + (size_t)sp.n_materials * 12);
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float4 *q0 = llights + (size_t)((sp.n_lights + 3) >> 2) * 6 + (size_t)wave * (3 * FQ_CAP + FQ_CAP / 4);
    float4 *q1 = q0 + FQ_CAP, *q2 = q1 + FQ_CAP;
    uint32_t *qp = reinterpret_cast<uint32_t *>(q2 + FQ_CAP);
    stage_material_lds(sp, lut, ldesc);
    stage_lights_lds(sp, llights);
    __syncthreads();
    const uint32_t xcd = blockIdx.x & 7u;
    uint32_t qn = 0, lit_px = 0;   // wave-uniform: records queued; lit pixels seen
    const uint32_t bpr = (sp.tiles_x + FT_BATCH - 1) / FT_BATCH;   // tickets per tile row: one ticket = FT_BATCH adjacent tiles
    for (uint32_t k = 0; k < 8; ++k) {
        const uint32_t src = (xcd + k) & 7u;
        if (src >= sp.tiles_y) continue;
        const uint32_t total = ((sp.tiles_y - 1 - src) / 8 + 1) * bpr;   // tickets of the rows y = src (mod 8)
        // a same-address atomic retires at ~88 per microsecond: FT_BATCH tiles per ticket keeps the eight counters far
        // below that, and the NEXT ticket is requested before the current batch is processed, so its round trip is hidden
        uint32_t t = 0;
        if (lane == 0) t = atomicAdd(sp.tickets + src * TICKET_STRIDE, 1u);
        t = __builtin_amdgcn_readfirstlane(t);
        while (t < total) {
            uint32_t t_next = 0;
            if (lane == 0) t_next = atomicAdd(sp.tickets + src * TICKET_STRIDE, 1u);
            const uint32_t ty = (t / bpr) * 8 + src, tx0 = (t % bpr) * FT_BATCH;
            for (uint32_t tx = tx0; tx < min(tx0 + FT_BATCH, sp.tiles_x); ++tx) {
                const TileHead cur = load_head(sp.g, ((size_t)ty * sp.tiles_x + tx) * 64 + lane);
                bool live;
                LitRec rec;
                const size_t gi = ((size_t)ty * sp.tiles_x + tx) * 64 + lane;
                const unsigned long long m = material_tile(sp, lut, ldesc, ty, tx, lane, cur, live, rec,
                                                           [&](float4 &gc, float4 &gd, float4 &ge) { gc = sp.g.c[gi]; gd = sp.g.d[gi]; ge = sp.g.e[gi]; });
                if (m == 0ull) continue;
                if (live) {
                    const uint32_t slot = qn + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                    q0[slot] = rec.r0; q1[slot] = rec.r1; q2[slot] = rec.r2; qp[slot] = rec.px;
                }
                const uint32_t add = (uint32_t)__popcll(m);
                qn += add; lit_px += add;
                if (qn >= 64) {   // pop the newest 64 (LDS operations of one wave execute in order: the records are there)
                    qn -= 64;
                    light_pixel<LIGHTS_PER_TRIP>(sp, llights, lane, q0[qn + lane], q1[qn + lane], q2[qn + lane], qp[qn + lane]);
                }
            }
            t = __builtin_amdgcn_readfirstlane(t_next);
        }
    }
    if (lane < qn) light_pixel<LIGHTS_PER_TRIP>(sp, llights, lane, q0[lane], q1[lane], q2[lane], qp[lane]);
    if (lane == 0) {
        if (sp.light_evals && lit_px) atomicAdd(sp.light_evals + 1, (unsigned long long)lit_px);
        const uint32_t done = atomicAdd(sp.tickets + 8 * TICKET_STRIDE, 1u);
        if (done == gridDim.x * 4 - 1) {   // every other wave has taken its last ticket: reset for the next pass
            for (uint32_t i = 0; i <= 8; ++i) sp.tickets[i * TICKET_STRIDE] = 0;
        }
    }
}

// PostProcessPass::run alone (post_process_pass.cpp:73-95): float RGBA in, RGBA8 (+ optional float rgb) out
__global__ __launch_bounds__(256) void k_post_process(const float4 *__restrict__ hdr, uint32_t n, int tm, float inv_gamma,
                                                      float exposure, uint32_t *__restrict__ rgba8, float *__restrict__ ldr) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float4 c = hdr[i];
    f3 l = post_process(mk(c.x, c.y, c.z), tm, inv_gamma, exposure);
    rgba8[i] = unorm8(l.x) | (unorm8(l.y) << 8) | (unorm8(l.z) << 16) | 0xFF000000u;
    if (ldr) { ldr[i * 3] = l.x; ldr[i * 3 + 1] = l.y; ldr[i * 3 + 2] = l.z; }
}

}  // namespace

// The shading pass = k_material over every tile, then k_light over the lit pixels it produced.
// n_bands > 1 cuts the frame into interleaved bands of 8-tile-row groups (band k = groups k, k + n_bands, ...) and
// pipelines them over two streams, k_light of band k beside k_material of band k + 1 (one memory bound, one VALU
// bound).  Measured on MI355X (4K, 64 lights): 1 band 0.44 ms, 2/4/8 bands 0.48-0.49 ms, 16 bands 0.56 ms -- the two
// grids do not co-reside usefully, with or without capping k_material's residency through LDS -- so the default is 1.
hipError_t launch_shade(const ShadeParams &sp0, const ShadeLaunch &L) {
    ShadeParams sp = sp0;
    uint32_t n_tiles = sp.tiles_x * sp.tiles_y;
    if (n_tiles == 0) return hipSuccess;
    const uint32_t bpr = (sp.tiles_x + 3) / 4, row_groups = (sp.tiles_y + 7) / 8;
    const uint32_t n_bands = std::max(1u, std::min(L.n_bands, row_groups));
    sp.n_bands = n_bands;
    hipError_t e = hipSuccess;
    const size_t lds_a = (256 + (size_t)sp.n_materials * 12) * sizeof(float);
    const size_t lds_b = std::max<size_t>(96, (size_t)((sp.n_lights + 3) / 4) * 96);
    if (L.fused && !L.from_vis) {
        sp.n_bands = 1; sp.band = 0;
        const size_t lds = fused_lds_bytes(sp.n_materials, sp.n_lights);
        if (L.mid && (e = hipEventRecord(L.mid, L.main)) != hipSuccess) return e;   // per-kernel timing: "k_material" part is empty
        if (L.lights_per_trip == 2) k_shade_fused<2><<<std::max(1u, L.fused_blocks), 256, lds, L.main>>>(sp);
        else k_shade_fused<4><<<std::max(1u, L.fused_blocks), 256, lds, L.main>>>(sp);
        return hipGetLastError();
    }
    for (uint32_t k = 0; k < n_bands; ++k) {
        sp.band = k;
        const uint32_t groups = (row_groups - k + n_bands - 1) / n_bands;
        if (L.inline_mode) {   // the material kernel finishes every pixel: no stream, no k_light
            const size_t lds = lds_a + (L.inline_mode == 2 ? lds_b : 0);
            if (L.from_vis && L.inline_mode == 2) k_material_vis<2><<<groups * 8 * bpr, 256, lds, L.main>>>(sp);
            else if (L.from_vis) k_material_vis<1><<<groups * 8 * bpr, 256, lds, L.main>>>(sp);
            else if (L.inline_mode == 2) k_material<2><<<groups * 8 * bpr, 256, lds, L.main>>>(sp);
            else k_material<1><<<groups * 8 * bpr, 256, lds, L.main>>>(sp);
            if ((e = hipGetLastError()) != hipSuccess) return e;
            if (L.mid && n_bands == 1 && (e = hipEventRecord(L.mid, L.main)) != hipSuccess) return e;
            continue;
        }
        if (L.from_vis) k_material_vis<0><<<groups * 8 * bpr, 256, lds_a, L.main>>>(sp);
        else k_material<0><<<groups * 8 * bpr, 256, lds_a, L.main>>>(sp);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        hipStream_t ls = L.main;
        if (L.mid && n_bands == 1 && (e = hipEventRecord(L.mid, L.main)) != hipSuccess) return e;
        if (n_bands > 1) {
            if ((e = hipEventRecord(L.band_done[k], L.main)) != hipSuccess) return e;
            if ((e = hipStreamWaitEvent(L.aux, L.band_done[k], 0)) != hipSuccess) return e;
            ls = L.aux;
        }
        if (L.lights_per_trip == 2) k_light<2><<<std::max(1u, std::min(L.light_blocks, groups * 8 * bpr)), 256, lds_b, ls>>>(sp);
        else k_light<4><<<std::max(1u, std::min(L.light_blocks, groups * 8 * bpr)), 256, lds_b, ls>>>(sp);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    if (n_bands > 1) {   // rejoin: later work on the main stream sees the finished frame
        if ((e = hipEventRecord(L.aux_done, L.aux)) != hipSuccess) return e;
        if ((e = hipStreamWaitEvent(L.main, L.aux_done, 0)) != hipSuccess) return e;
    }
    return hipSuccess;
}

size_t fused_lds_bytes(uint32_t n_materials, uint32_t n_lights) {
    return (256 + (size_t)n_materials * 12) * sizeof(float) + (size_t)((n_lights + 3) / 4) * 96 + 4 * (size_t)(3 * FQ_CAP + FQ_CAP / 4) * 16;
}

// resident workgroups per CU of the fused kernel for this LDS size (the persistent grid is CUs x this)
int fused_blocks_per_cu(size_t lds_bytes, uint32_t lights_per_trip) {
    int n = 0;
    hipError_t e = lights_per_trip == 2 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_shade_fused<2>, 256, lds_bytes)
                                        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_shade_fused<4>, 256, lds_bytes);
    return e == hipSuccess ? n : 0;
}

hipError_t launch_post_process(const float4 *hdr, uint32_t w, uint32_t h, int32_t tm, float inv_gamma, float exposure,
                               uint8_t *rgba8, float *ldr, hipStream_t s) {
    uint32_t n = w * h;
    if (n == 0) return hipSuccess;
    k_post_process<<<(n + 255) / 256, 256, 0, s>>>(hdr, n, tm, inv_gamma, exposure, reinterpret_cast<uint32_t *>(rgba8), ldr);
    return hipGetLastError();
}

}  // namespace arctic
