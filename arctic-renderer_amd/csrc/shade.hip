// shade.hip -- the hot path: per-pixel forward PBR shading + fused tonemap for gfx950.
//
// Replaces ps_main of shaders/forward.hlsl:208-235 (material fetch :98-124, calculate_shadow
// :68-96, Cook-Torrance GGX :126-193, point-light loop :224-231) and main of
// shaders/post_process.hlsl:59-93 (Reinhard / exposure / ACES :39-57, gamma :34-37) with ONE
// HIP kernel that reads the tile-major G-buffer written by geometry.hip and stores RGBA8.
//
// Mapping: 1 wavefront = one 8x8 pixel tile (lane l = pixel (l&7, l>>3)), 1 workgroup = 4 waves =
// 16x16 pixels.  G-buffer reads are 16 B / lane, contiguous per wave.  There is no texture
// hardware on gfx950, so every Sample() is address arithmetic + global_load_dword + weights;
// sRGB decode is a 256-entry LDS table.  Lights are wave-uniform, so they are fetched with scalar
// loads and live in SGPRs.  Scalar-per-pixel FP32 (VALU): no MFMA, by design.
//
// Numerics: texel coordinates/weights and the whole shadow test are computed exactly as the
// CPU oracle does (fp contract off, IEEE divide) because they feed discontinuous decisions; the
// BRDF and tonemap use v_rcp/v_rsq/v_exp/v_log (~1 ulp) and free contraction, inside the 1e-4
// per-channel budget of the output.
#include "common.h"

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

constexpr float PI = 3.14159265f;   // forward.hlsl:1
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

__device__ __forceinline__ Taps fetch_taps(const TexDesc &d, float u, float v) {
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
struct Surface { f3 base, n; float metal, rough; };

__device__ __forceinline__ f3 outgoing_radiance(const Surface &s, f3 wo, f3 wi, f3 Li) {
    f3 h = normalize(wo + wi);
    float ndwo = fmaxf(dot(s.n, wo), 0.0f), ndwi = fmaxf(dot(s.n, wi), 0.0f);
    float ndh = fmaxf(dot(s.n, h), 0.0f), hdwo = fmaxf(dot(h, wo), 0.0f);
    // fresnel_schlick :126-129
    f3 F0 = mk(0.04f + s.metal * (s.base.x - 0.04f), 0.04f + s.metal * (s.base.y - 0.04f), 0.04f + s.metal * (s.base.z - 0.04f));
    float m = sat(1.0f - hdwo), m2 = m * m, p5 = m2 * m2 * m;
    f3 F = mk(F0.x + (1.0f - F0.x) * p5, F0.y + (1.0f - F0.y) * p5, F0.z + (1.0f - F0.z) * p5);
    // distribution_ggx :131-143
    float a = s.rough * s.rough, a2 = a * a;
    float d = ndh * ndh * (a2 - 1.0f) + 1.0f;
    float NDF = a2 * rcp(PI * d * d);
    // geometry_smith :145-163
    float r1 = s.rough + 1.0f, k = r1 * r1 * 0.125f;
    float G = ndwo * rcp(ndwo * (1.0f - k) + k) * ndwi * rcp(ndwi * (1.0f - k) + k);
    // brdf_cook_torrance :165-175
    float spec = NDF * G * rcp(4.0f * ndwo * ndwi + 0.0001f);
    // calculate_outgoing_radiance :177-193
    float km = 1.0f - s.metal;
    f3 kD = mk((1.0f - F.x) * km, (1.0f - F.y) * km, (1.0f - F.z) * km);
    f3 c = mk(kD.x * s.base.x * INV_PI + spec * F.x, kD.y * s.base.y * INV_PI + spec * F.y, kD.z * s.base.z * INV_PI + spec * F.z);
    return c * Li * ndwi;
}

// ---- post_process.hlsl ---------------------------------------------------------------------------
__device__ __forceinline__ float rrt_odt(float c) {
    float a = c * (c + 0.0245786f) - 0.000090537f;
    float b = c * (0.983729f * c + 0.4329510f) + 0.238081f;
    return a * rcp(b);
}
__device__ __forceinline__ f3 post_process(f3 c, int tm, float inv_gamma, float exposure) {
    f3 t;
    if (tm == 1) {          // tm_exposure :44-47
        const float LOG2E = 1.4426950408889634f;
        t = mk(1.0f - __builtin_amdgcn_exp2f(-c.x * exposure * LOG2E), 1.0f - __builtin_amdgcn_exp2f(-c.y * exposure * LOG2E),
               1.0f - __builtin_amdgcn_exp2f(-c.z * exposure * LOG2E));
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

__global__ __launch_bounds__(256) void k_shade(const ShadeParams sp) {
    __shared__ float lut[256];
    lut[threadIdx.x] = sp.srgb_lut[threadIdx.x];
    __syncthreads();

    const uint32_t bw = (sp.tiles_x + 1) >> 1;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t tx = (blockIdx.x % bw) * 2 + (wave & 1), ty = (blockIdx.x / bw) * 2 + (wave >> 1);
    if (tx >= sp.tiles_x || ty >= sp.tiles_y) return;
    const size_t idx = ((size_t)ty * sp.tiles_x + tx) * 64 + lane;
    const uint32_t x = tx * 8 + (lane & 7);
    const int32_t y = (int32_t)(ty * 8 + (lane >> 3)) - (int32_t)sp.row0_in_tile;
    const bool in_frame = x < sp.width && y >= 0 && y < (int32_t)sp.rows;

    const float4 q0 = sp.g.p0[idx], q1 = sp.g.p1[idx], q2 = sp.g.p2[idx], q3 = sp.g.p3[idx];
    const float nx = sp.g.p4[idx * 3], ny = sp.g.p4[idx * 3 + 1], nz = sp.g.p4[idx * 3 + 2];
    const uint32_t mat = __float_as_uint(q0.w);
    const bool covered = in_frame && mat < sp.n_materials;

    f3 color = mk(0.0f, 0.0f, 0.0f);
    if (covered) {
        // ---- material fetch, forward.hlsl:98-124 ------------------------------------------------
        const TexDesc *td = sp.tex + (size_t)mat * 3;
        const float u = q2.x, v = q2.y;
        Surface s;
        {
            Taps t = fetch_taps(td[0], u, v);
            s.base = mk(filt_srgb(t, 0, lut), filt_srgb(t, 1, lut), filt_srgb(t, 2, lut));
        }
        {
            Taps t = fetch_taps(td[1], u, v);
            float r = filt_unorm(t, 0), g = 1.0f - filt_unorm(t, 1), b = filt_unorm(t, 2);   // normal.g = 1 - normal.g
            r = r * 2.0f - 1.0f; g = g * 2.0f - 1.0f; b = b * 2.0f - 1.0f;
            // mul(tbn, v), tbn columns t, b, n
            f3 T = mk(q2.z, q2.w, q3.x), B = mk(q3.y, q3.z, q3.w), N = mk(nx, ny, nz);
            s.n = normalize(T * r + B * g + N * b);
        }
        {
            Taps t = fetch_taps(td[2], u, v);
            s.rough = filt_unorm(t, 1);   // .g
            s.metal = filt_unorm(t, 2);   // .b
        }
        const f3 world = mk(q0.x, q0.y, q0.z);
        const f3 wo = normalize(mk(sp.eye[0], sp.eye[1], sp.eye[2]) - world);
        const float shadow = calculate_shadow(sp.shadow_map, sp.shadow_size, q1);
        const float lit = 1.0f - shadow;
        f3 Lo = mk(0.0f, 0.0f, 0.0f);
        // exact culling: everything below is multiplied by (1 - shadow) (point lights too: forward.hlsl:230),
        // so a wave whose pixels are all fully shadowed skips the sun and the whole light loop
        const bool wave_lit = !sp.culling || __ballot(lit != 0.0f) != 0ull;
        if (wave_lit) {
            Lo = outgoing_radiance(s, wo, mk(-sp.sun_dir[0], -sp.sun_dir[1], -sp.sun_dir[2]),
                                   mk(sp.sun_color[0], sp.sun_color[1], sp.sun_color[2])) * lit;
            for (uint32_t i = 0; i < sp.n_lights; ++i) {
                const float4 lp = sp.lights[2 * i], lc = sp.lights[2 * i + 1];
                const f3 d = mk(lp.x, lp.y, lp.z) - world;
                // exact culling: n.wi <= 0 zeroes the light (forward.hlsl:191-192); skip it when that holds wave-wide
                if (sp.culling && __ballot(lit != 0.0f && dot(s.n, d) > 0.0f) == 0ull) continue;
                const float d2 = dot(d, d), inv = rsq(d2);
                const f3 wi = d * inv;
                const f3 radiance = mk(lc.x, lc.y, lc.z) * (inv * inv);
                Lo = Lo + outgoing_radiance(s, wo, wi, radiance) * lit;
                if (sp.light_evals) {
                    const unsigned long long active = __ballot(1);
                    if (lane == (uint32_t)__ffsll((long long)active) - 1) atomicAdd(sp.light_evals, (unsigned long long)__popcll(active));
                }
            }
        }
        color = Lo + s.base * sp.ambient;
    }
    if (!in_frame) return;
    const f3 l = post_process(color, sp.tm_method, sp.inv_gamma, sp.exposure);
    const size_t p = (size_t)y * sp.width + x;
    reinterpret_cast<uint32_t *>(sp.out_rgba8)[p] = unorm8(l.x) | (unorm8(l.y) << 8) | (unorm8(l.z) << 16) | 0xFF000000u;
    if (sp.out_ldr) { sp.out_ldr[p * 3] = l.x; sp.out_ldr[p * 3 + 1] = l.y; sp.out_ldr[p * 3 + 2] = l.z; }
    if (sp.out_hdr) { sp.out_hdr[p * 3] = color.x; sp.out_hdr[p * 3 + 1] = color.y; sp.out_hdr[p * 3 + 2] = color.z; }
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

hipError_t launch_shade(const ShadeParams &sp, hipStream_t s) {
    uint32_t bw = (sp.tiles_x + 1) / 2, bh = (sp.tiles_y + 1) / 2;
    if (bw * bh == 0) return hipSuccess;
    k_shade<<<bw * bh, 256, 0, s>>>(sp);
    return hipGetLastError();
}

hipError_t launch_post_process(const float4 *hdr, uint32_t w, uint32_t h, int32_t tm, float inv_gamma, float exposure,
                               uint8_t *rgba8, float *ldr, hipStream_t s) {
    uint32_t n = w * h;
    if (n == 0) return hipSuccess;
    k_post_process<<<(n + 255) / 256, 256, 0, s>>>(hdr, n, tm, inv_gamma, exposure, reinterpret_cast<uint32_t *>(rgba8), ldr);
    return hipGetLastError();
}

}  // namespace arctic
