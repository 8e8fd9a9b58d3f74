// shade.hip -- the hot path: per-pixel forward PBR shading + fused tonemap for gfx950.
//
// Replaces ps_main of shaders/forward.hlsl:208-235 (material fetch :98-124, calculate_shadow
// :68-96, Cook-Torrance GGX :126-193, point-light loop :224-231) and main of
// shaders/post_process.hlsl:59-93 (Reinhard / exposure / ACES :39-57, gamma :34-37), plus the skybox lookup of
// shaders/skybox.hlsl:61-90 for pixels without geometry; reads the tile-major G-buffer written by geometry.hip (or, for
// whole frames, the visibility plane: k_material_vis) and stores RGBA8.
//
// ONE kernel per pass.  1 wavefront = one 8x8 screen tile (lane l = pixel (l&7, l>>3)), 4 tiles per workgroup:
//   A  head of the G-buffer (uv, light-space position, material: 28 B/pixel, 16 B/lane contiguous loads); the material is
//      wave-uniform in almost every tile, so its descriptor comes through the SCALAR unit (readlane + s_load, a waterfall
//      loop over the distinct materials of a mixed tile) and the four 8-byte texel loads of the bilinear footprint use a
//      scalar base + 32-bit lane offsets;
//   B  shadow test: a conservative min/max table of the shadow map (k_shadow_bounds: one entry per 4x4-aligned 8x8 texel
//      block) decides most pixels with ONE 8-byte load -- exactly, because a bilinear tap cannot leave the interval of its
//      texels; only tiles on a shadow edge load the 4x4 window and run the 25 compares;
//   C  fully shadowed pixels are finished (ambient * base: every light term of ps_main carries (1 - shadow),
//      forward.hlsl:222,230) without ever loading their position / tangent frame (48 B/pixel);
//   D  lit pixels: normal map, sun + point lights.  Two loops, both fed through the scalar cache: plain fp32 (few lights), or two
//      lights at a time in packed fp32 with SGPR light pairs and op_sel-shared per-pixel register pairs (many lights);
//   E  one tonemap + gamma + RGBA8 store for every pixel of the tile.
// Scalar-per-pixel FP32: no MFMA, by design.  The kernel is VALU-issue bound with many lights and memory bound with few:
// it is written against the issue costs measured by tools/experiments/valu_rates.hip (profiles/r2_valu_rates.txt):
// v_add/mul/fma ~2.6-2.9 cycles per wave64, v_pk_* ~4.4, min/max/cvt/cmp/floor/bfe ~4.2, transcendentals 8.
//
// Numerics: texel coordinates/weights and the whole shadow test are computed exactly as the
// CPU oracle does (IEEE divide, every operation rounding once in the oracle's order) because they feed
// discontinuous decisions; the BRDF and tonemap use v_rcp/v_rsq/v_exp/v_log (~1 ulp), inside the 1e-4
// per-channel budget of the output.  The file is compiled with -ffp-contract=off and every fused
// multiply-add is written out, so k_material and k_material_vis round identically (their images are
// compared bit for bit).
#include "common.h"
#include "edges.h"
#include "shadow_coords.h"

#ifndef ARCTIC_WG_WAVES
#define ARCTIC_WG_WAVES 1       // waves per workgroup of k_material, the pass over a G-buffer: 1 (a tile per workgroup; round 5) or 4 (a strip of 4 tiles per workgroup: rounds 1-4).
#endif                          // A workgroup's slot is held until its LAST wave ends; with tiles of unequal length (shadowed 3 us, lit 13) one-wave workgroups keep
                                // 5.1 instead of 4.5 waves resident per SIMD.  Round 4 measured them faster with few lights and 1.8 % slower at 64 (with its dispatch order on);
                                // on round 5's build, order off: 64 / 16 / 0 lights 0.1874 / 0.1273 / 0.0951 -> 0.1857 / 0.1224 / 0.0936 ms (profiles/r5_l_ab_one_wave_workgroups.txt).
                                // k_material_vis keeps 4 (VIS_WG_WAVES): whole frames are 9 % slower with one.
#ifndef ARCTIC_VIS_WG_WAVES
#define ARCTIC_VIS_WG_WAVES 4
#endif
constexpr int VIS_WG_WAVES = ARCTIC_VIS_WG_WAVES;   // (2 and 1 measured: whole frames 0.2765 -> 0.2955 / 0.2964 ms, profiles/r5_l_ab_one_wave_workgroups.txt)
#ifndef ARCTIC_LUT_SHARED
#define ARCTIC_LUT_SHARED 0     // A/B switch: 1 = each wave of a workgroup loads a quarter of the sRGB table (256 B instead of 1 KiB per wave) and a barrier stands behind the stores
#endif
#ifndef ARCTIC_PCF_CANDIDATES
#define ARCTIC_PCF_CANDIDATES 0   // A/B switch: 1 = the 25 vertical lerps of k_material's PCF taps evaluate three candidate cells and select the RESULT instead of selecting
                                  // the operands of every lerp (same bits; measured -0.4 % at 0 / 16 lights, nothing at 64: profiles/r5_b_ab_inplace_sums_and_pcf_candidates.txt --
                                  // and 74 VGPRs once the kernel also carries the D3D-style sampler's taps, a wave per SIMD: off)
#endif
#ifndef ARCTIC_PCF_ROW_CANDIDATES
#define ARCTIC_PCF_ROW_CANDIDATES 0   // A/B switch: 1 = the horizontal lerps of the 25 PCF taps as candidates too (register pressure: see shadow_window)
#endif
#ifndef ARCTIC_PIN_SECOND_WAVE
#define ARCTIC_PIN_SECOND_WAVE 1   // A/B switch: see shade_tile_fast
#endif
#ifndef ARCTIC_TILED_FETCH
#define ARCTIC_TILED_FETCH 1   // A/B switch (instruction census only): 0 = no code for tiled material images (such materials then render wrongly)
#endif
#ifndef ARCTIC_EDGE_IN_FAST
#define ARCTIC_EDGE_IN_FAST 1   // A/B switch (build_tmp variants only): 0 = a tile on a shadow edge goes to the general tile, as in round 3
#endif

namespace arctic {

namespace {

// The kernels read their argument block through the CONSTANT address space at the point of use (scalar loads).  k_material
// shades up to two tiles one after the other in a loop: there the pointer to the block and the lane index pass through an empty asm
// once per tile, so that nothing derived from them can be hoisted in front of the loop and kept in registers through it (the
// block is 100 dwords: left alone the compiler loads it once, spills SGPRs into VGPR lanes and keeps per-lane addresses alive).
typedef const ShadeParams __attribute__((address_space(4))) &SP;
typedef const ShadeParams __attribute__((address_space(4))) *KernArgs;

// The argument block in BATCHES (round 4).  Read field by field at the point of use, a tile's wave paid a scalar-cache round trip per
// field -- some twenty of them one after the other, five between the arrival of the tile's first bytes and the texel loads that depend
// on them.  ShadeParams is laid out in the order a tile needs its fields (common.h), each phase reads its block into registers at ONE
// place, behind the vector loads of the phase before, and an empty asm statement that "modifies" all of them pins them there: the
// compiler merges the adjacent scalar loads and waits once, in the shadow of memory traffic that is in flight anyway.
struct ArgsA { const float4 *ga; const float *gb; uint32_t tiles_x, tiles_y, T, stride; int32_t debug; uint32_t n_materials; };
__device__ __forceinline__ ArgsA args_a(KernArgs a) {   // before a tile's first bytes can be asked for
    ArgsA r = {a->g.a, a->g.b, a->tiles_x, a->tiles_y, a->tiles_per_wave, a->group_stride, a->debug, a->n_materials};
    asm volatile("" : "+s"(r.ga), "+s"(r.gb), "+s"(r.tiles_x), "+s"(r.tiles_y), "+s"(r.T), "+s"(r.stride), "+s"(r.debug), "+s"(r.n_materials));
    return r;
}
// ... and, for a wave's first tile, more in the same batch: the sRGB table, the visibility plane (k_material_vis), the dispatch order
typedef const uint32_t __attribute__((address_space(4))) *OrderList;   // (read with scalar loads)
struct OrderArgs { OrderList order; uint32_t n_jobs; };
__device__ __forceinline__ ArgsA args_a_first(KernArgs a, const float *&srgb_lut, const unsigned long long *&vis, OrderArgs &O) {
    ArgsA r = {a->g.a, a->g.b, a->tiles_x, a->tiles_y, a->tiles_per_wave, a->group_stride, a->debug, a->n_materials};
    srgb_lut = a->srgb_lut; vis = a->vis;
    const uint32_t *order = a->tile_order; O.n_jobs = a->n_jobs;
    asm volatile("" : "+s"(r.ga), "+s"(r.gb), "+s"(r.tiles_x), "+s"(r.tiles_y), "+s"(r.T), "+s"(r.stride), "+s"(r.debug), "+s"(r.n_materials), "+s"(srgb_lut), "+s"(vis),
                 "+s"(order), "+s"(O.n_jobs));
    O.order = (OrderList)order;
    return r;
}
__device__ __forceinline__ OrderArgs order_args(KernArgs a) {
    const uint32_t *order = a->tile_order; uint32_t n = a->n_jobs;
    asm volatile("" : "+s"(order), "+s"(n));
    OrderArgs O = {(OrderList)order, n};
    return O;
}
constexpr int32_t DEBUG_TRACE = 1 << 30;   // ShadeParams::debug: the host asked for a tile trace (ShadeParams::trace is set)
// ShadeParams::debug bits 20..22 = ARCTIC_OPT_SAMPLER: the D3D-style sampler variants the oracle has had since round 4 (oracle/arctic_oracle.cpp SAMPLER_*),
// now selectable on the HIP side.  The reference's filter arithmetic is its D3D12 sampler's (forward_pass.cpp:38-51, MIN_MAG_MIP_LINEAR + WRAP); D3D lets the
// hardware keep texel coordinates in fixed point with 8 fractional bits (D3D11.3 functional specification 3.2.4.1 / 7.18.8):
//   bit 0  material textures: the scaled coordinate u W - 0.5 snapped to 1/256 texel (round to nearest) before the index / weight split
//   bit 2  the same for the 25 PCF taps of the shadow map (forward.hlsl:84-92 goes through the same sampler)
// (bit 1, sRGB decoded after filtering, stays an oracle-only bound: a conformant D3D10+ sampler decodes first.)  Wave-uniform branches: the default costs nothing.
constexpr int SAMPLER_SHIFT = 20;
constexpr int32_t SAMPLER_Q8_MATERIAL = 1 << SAMPLER_SHIFT, SAMPLER_Q8_SHADOW = 4 << SAMPLER_SHIFT;
struct ShadowArgs { const float *map; const float2 *bounds; uint32_t S, pitch; };   // calculate_shadow's inputs
struct ArgsB { const TexDesc *tex; ShadowArgs sh; uint8_t *out; uint32_t width, rows, row0_in_tile; int32_t culling, hdr16, tm; };
__device__ __forceinline__ ArgsB args_b(KernArgs a) {   // while the head of the tile is in flight
    ArgsB r = {a->tex, {a->shadow_map, a->shadow_bounds, a->shadow_size, a->bounds_pitch}, a->out_rgba8, a->width, a->rows, a->row0_in_tile, a->culling, a->hdr16, a->tm_method};
    asm volatile("" : "+s"(r.tex), "+s"(r.sh.map), "+s"(r.sh.bounds), "+s"(r.sh.S), "+s"(r.sh.pitch), "+s"(r.out), "+s"(r.width), "+s"(r.rows),
                 "+s"(r.row0_in_tile), "+s"(r.culling), "+s"(r.hdr16), "+s"(r.tm));
    return r;
}
// what post_process + the store need (store_pixel)
struct StoreArgs { int32_t hdr16, debug, tm; float exposure, inv_gamma; float *out_ldr, *out_hdr; };
struct ArgsC { float ambient; StoreArgs st; const float4 *gc, *gd, *ge; };
__device__ __forceinline__ ArgsC args_c(KernArgs a, const ArgsA &A, const ArgsB &B) {   // while the texels and the shadow-table entry are in flight
    ArgsC r = {a->ambient, {B.hdr16, A.debug, B.tm, a->exposure, a->inv_gamma, a->out_ldr, a->out_hdr}, a->g.c, a->g.d, a->g.e};
    asm volatile("" : "+s"(r.ambient), "+s"(r.st.exposure), "+s"(r.st.inv_gamma), "+s"(r.st.out_ldr), "+s"(r.st.out_hdr), "+s"(r.gc), "+s"(r.gd), "+s"(r.ge));
    return r;
}
// what the fast tile's epilogue needs (ambient term, post_process, the store), as ONE set of scalar registers: taken from the batches
// above for a tile without a lit pixel, loaded again BEHIND the light loop for a tile with one -- so that none of it occupies scalar
// registers through the loop.  (The SIMD's scalar register file decides the kernel's occupancy as much as the vector one does:
// tools/experiments/occupancy2.hip -- 7 waves need <= 96 SGPRs, 8 waves <= 80, whatever the VGPR count allows.)
struct EpiArgs { StoreArgs st; uint8_t *out; uint32_t width, row0_in_tile; float ambient; };
__device__ __forceinline__ EpiArgs epi_args(KernArgs a) {
    EpiArgs r = {{a->hdr16, a->debug, a->tm_method, a->exposure, a->inv_gamma, a->out_ldr, a->out_hdr}, a->out_rgba8, a->width, a->row0_in_tile, a->ambient};
    asm volatile("" : "+s"(r.st.hdr16), "+s"(r.st.debug), "+s"(r.st.tm), "+s"(r.st.exposure), "+s"(r.st.inv_gamma), "+s"(r.st.out_ldr), "+s"(r.st.out_hdr), "+s"(r.out),
                 "+s"(r.width), "+s"(r.row0_in_tile), "+s"(r.ambient));
    return r;
}
__device__ __forceinline__ ShadowArgs shadow_args(SP sp) { ShadowArgs r = {sp.shadow_map, sp.shadow_bounds, sp.shadow_size, sp.bounds_pitch}; return r; }
__device__ __forceinline__ StoreArgs store_args(SP sp) { StoreArgs r = {sp.hdr16, sp.debug, sp.tm_method, sp.exposure, sp.inv_gamma, sp.out_ldr, sp.out_hdr}; return r; }
// the lit pixels' constants: eye and the light list (in front of the light loop), the sun (where it is evaluated: behind the packed loop)
struct LightArgs { float eye[3]; uint32_t n_lights; const float4 *pairs, *lights; };
__device__ __forceinline__ LightArgs light_args(SP sp) {
    LightArgs r = {{sp.eye[0], sp.eye[1], sp.eye[2]}, sp.n_lights, sp.light_pairs, sp.lights};
    asm volatile("" : "+s"(r.eye[0]), "+s"(r.eye[1]), "+s"(r.eye[2]), "+s"(r.n_lights), "+s"(r.pairs), "+s"(r.lights));
    return r;
}
struct SunArgs { float dir[3], color[3]; };
__device__ __forceinline__ SunArgs sun_args(SP sp) {
    SunArgs r = {{sp.sun_dir[0], sp.sun_dir[1], sp.sun_dir[2]}, {sp.sun_color[0], sp.sun_color[1], sp.sun_color[2]}};
    asm volatile("" : "+s"(r.dir[0]), "+s"(r.dir[1]), "+s"(r.dir[2]), "+s"(r.color[0]), "+s"(r.color[1]), "+s"(r.color[2]));
    return r;
}

struct f3 { float x, y, z; };
__device__ __forceinline__ f3 mk(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 operator*(f3 a, f3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
// This file is compiled with -ffp-contract=off: every fused multiply-add is written out (fm), so the shared device
// functions round identically in every kernel they are inlined into (G-buffer pass and visibility-plane pass agree bit for bit).
__device__ __forceinline__ float fm(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ float dot(f3 a, f3 b) { return fm(a.z, b.z, fm(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ f3 normalize(f3 v) { return v * rsq(dot(v, v)); }
__device__ __forceinline__ float sat(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, 1.0f); }
__device__ __forceinline__ float pow_fast(float x, float e) {   // x >= 0
    return __builtin_amdgcn_exp2f(e * __builtin_amdgcn_logf(x));
}

// the lane's index in its wave, computed afresh wherever it is asked for (two instructions; volatile: never merged with an earlier one):
// a lane index kept in a register lives through the light loop just to address the store behind it
__device__ __forceinline__ uint32_t wave_lane() { uint32_t l; asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l)); return l; }

// PI = 3.14159265 as written at forward.hlsl:1
constexpr float INV_PI = 1.0f / 3.14159265f;

// ---- sampler: MIN_MAG_MIP_LINEAR + WRAP (forward_pass.cpp:38-51), texel centres at +0.5 ----------
// exact arithmetic: these coordinates select texels and weights
// 24.8 fixed point, round to nearest: the oracle's floor(x * 256 + 0.5) / 256 (x * 256 and / 256 are exact)
__device__ __forceinline__ float snap256(float x) {
#pragma clang fp contract(off)
    return floorf(x * 256.0f + 0.5f) * 0.00390625f;
}
__device__ __forceinline__ void wrap_axis(float u, uint32_t n, int &i0, int &i1, float &f, bool q8 = false /* wave-uniform */) {
#pragma clang fp contract(off)
    float uw = u - floorf(u);
    float x = uw * (float)n - 0.5f;
    if (q8) x = snap256(x);
    float xf = floorf(x);
    f = x - xf;
    i0 = (int)xf;                       // in [-1, n - 1]
    i1 = i0 + 1;                        // in [0, n]
    i0 = i0 < 0 ? (int)n - 1 : i0;      // WRAP: one compare + select each (uw in [0, 1] keeps the indices within one period)
    i1 = i1 == (int)n ? 0 : i1;
}

// one texture of a material, WAVE-UNIFORM (the descriptor lives in SGPRs).  packed = 0: plain RGBA8 image.  packed = 1: the
// material's three equally sized images are stored as ONE image of 8-byte texels holding exactly the eight channels ps_main
// reads (forward.hlsl:98-124), with a one-texel WRAP border around it (common.h TexDesc):
//   word 0 = diffuse.r | diffuse.g << 8 | diffuse.b << 16 | normal.r << 24
//   word 1 = normal.g | normal.b << 8 | metal_rough.g << 16 | metal_rough.b << 24
struct TexS { const uint8_t *texels; uint32_t w, h, packed; float wf, hf; uint32_t pitch, tile_row_bytes; };
// The descriptor is fetched with an explicit s_load: written as a plain load the compiler sinks it into the `mat == m` branch
// of the waterfall loop, replaces the uniform m by the per-lane mat it equals there, and issues a vector load per lane.
typedef uint32_t u4v __attribute__((ext_vector_type(4)));
typedef uint32_t u8v __attribute__((ext_vector_type(8)));
__device__ __forceinline__ TexS tex_decode(const u8v &v) {
    TexS t;
    t.texels = reinterpret_cast<const uint8_t *>(((unsigned long long)v[1] << 32) | v[0]);
    t.w = v[2] & 0x7FFFFFFFu; t.h = v[3];
    t.packed = v[2] >> 31;   // TexDesc::w bit 31
    t.wf = __uint_as_float(v[4]); t.hf = __uint_as_float(v[5]); t.pitch = v[6];
    // (readfirstlane of a value that sits in an SGPR is a scalar move -- and tells the compiler's uniformity analysis so: the asm statements the descriptor
    // comes through also carry a vector operand, which makes all their results "divergent" on paper; a branch on such a value turned the epilogue's scalar
    // address arithmetic into vector instructions, 13 per tile -- round 5, found with SQ_INSTS_VALU)
    t.tile_row_bytes = __builtin_amdgcn_readfirstlane(v[7]);
    return t;
}
__device__ __forceinline__ TexS tex_desc(const TexDesc *tex, uint32_t i /* wave-uniform */) {
    u8v v;
    asm("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(tex + i));   // not volatile: a side-effecting asm would stop the compiler from using scalar loads for the lights
    return tex_decode(v);
}
// The same in two steps, for the fast tile: the descriptor's load goes out as soon as the material is known and is waited for only
// behind the shadow-coordinate arithmetic (some twenty vector instructions that do not need it), instead of a scalar-cache round trip
// with nothing to do.  `before` / `after`: a value the work in between starts from / ends with, passed through the two statements so
// that data flow, not the scheduler's mood, keeps that work between them.  tools/isa_lint.py checks that nothing touches the
// destination registers before the wait.
__device__ __forceinline__ void tex_desc_issue(const TexDesc *tex, uint32_t i /* wave-uniform */, u8v &v, float &before) {
    asm volatile("s_load_dwordx8 %0, %2, 0x0" : "=&s"(v), "+v"(before) : "s"(tex + i));
}
__device__ __forceinline__ TexS tex_desc_wait(u8v &v, uint32_t &after) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(v), "+v"(after));
    return tex_decode(v);
}

// load through a WAVE-UNIFORM base pointer (SGPR pair) and a 32-bit per-lane byte offset: one global_load with the saddr
// form, no 64-bit address arithmetic per lane.  (Pointers rebuilt from descriptor words would otherwise be generic: flat_load.)
typedef const char __attribute__((address_space(1))) *gchar;
typedef uint32_t u2v __attribute__((ext_vector_type(2)));
typedef uint32_t u4u __attribute__((ext_vector_type(4), aligned(8)));    // 16 bytes at an 8-byte aligned address: gfx950 runs in unaligned-access mode
typedef float f2v __attribute__((ext_vector_type(2)));
typedef float float4u __attribute__((ext_vector_type(4), aligned(4)));   // 16 bytes at a 4-byte aligned address
__device__ __forceinline__ uint32_t gload_u32(const void *base, uint32_t o) { return *(const uint32_t __attribute__((address_space(1))) *)((gchar)base + o); }
__device__ __forceinline__ uint2 gload_u2(const void *base, uint32_t o) { const u2v v = *(const u2v __attribute__((address_space(1))) *)((gchar)base + o); return make_uint2(v.x, v.y); }
__device__ __forceinline__ u4v gload_u4u(const void *base, uint32_t o) { const u4u v = *(const u4u __attribute__((address_space(1))) *)((gchar)base + o); return (u4v){v.x, v.y, v.z, v.w}; }
__device__ __forceinline__ float2 gload_f2(const void *base, uint32_t o) { const f2v v = *(const f2v __attribute__((address_space(1))) *)((gchar)base + o); return make_float2(v.x, v.y); }
__device__ __forceinline__ float4u gload_f4u(const void *base, uint32_t o) { return *(const float4u __attribute__((address_space(1))) *)((gchar)base + o); }

// texel coordinate along one axis WITHOUT the wrap (the bordered packed images never need it): the oracle's operations in the
// oracle's order, i0 = first texel of the footprint in [-1, n - 1], f = weight of the second
__device__ __forceinline__ float axis_scaled(float u, float nf) {
#pragma clang fp contract(off)
    const float uw = u - floorf(u);
    return uw * nf - 0.5f;
}
__device__ __forceinline__ void axis_split(float x, int &i0, float &f) {
#pragma clang fp contract(off)
    const float xf = floorf(x);
    f = x - xf;
    i0 = (int)xf;
}
__device__ __forceinline__ void axis_nowrap(float u, float nf, int &i0, float &f) {
#pragma clang fp contract(off)
    const float uw = u - floorf(u);
    const float x = uw * nf - 0.5f;
    const float xf = floorf(x);
    f = x - xf;
    i0 = (int)xf;
}

// bilinear footprint of a lane: four texels (8 bytes each when packed, else 4) + weights.  Byte offsets are 32-bit (images
// hold at most 2^29 texels), so a load is one global_load with the descriptor's base in SGPRs.
// r0 = {texel (x0, y0) word 0, word 1, texel (x1, y0) word 0, word 1}, r1 = the same of row y1 (plain images: words 0 only)
struct Taps { u4v r0, r1; float w00, w10, w01, w11; };
__device__ __forceinline__ void tap_weights(float fx, float fy, Taps &t) {
    const float gx = 1.0f - fx, gy = 1.0f - fy;
    t.w00 = gx * gy; t.w10 = fx * gy; t.w01 = gx * fy; t.w11 = fx * fy;
}
// packed + bordered image: the footprint is two 16-byte loads, their address one multiply-add and one shifted add
__device__ __forceinline__ void fetch_taps_packed(const TexS &d, float u, float v, Taps &t, bool q8 = false /* wave-uniform: ARCTIC_OPT_SAMPLER bit 0 */) {
    int x0, y0;
    float fx, fy;
    float x = axis_scaled(u, d.wf), y = axis_scaled(v, d.hf);
    if (q8) { asm volatile(""); x = snap256(x); y = snap256(y); }   // (a real branch: one for both axes, nothing for the default sampler)
    axis_split(x, x0, fx);
    axis_split(y, y0, fy);
    if (ARCTIC_TILED_FETCH && d.tile_row_bytes) {   // (wave-uniform) 4 x 4-texel tiles, common.h TexDesc::tile_row_bytes: four 8-byte loads, a texel of the footprint may sit in the next tile
        asm volatile("");
        const uint32_t X = (uint32_t)(x0 + 1), Y = (uint32_t)(y0 + 1), rx = X & 3u, ry = Y & 3u;
        const uint32_t o00 = ((__umul24(Y >> 2, d.tile_row_bytes >> 7) + (X >> 2)) << 7) + ((ry * 4u + rx) << 3);
        const uint32_t dx = rx == 3u ? 128u - 24u : 8u, dy = ry == 3u ? d.tile_row_bytes - 96u : 32u;
        const uint2 a = gload_u2(d.texels, o00), b = gload_u2(d.texels, o00 + dx), c = gload_u2(d.texels, o00 + dy), e = gload_u2(d.texels, o00 + dy + dx);
        t.r0 = (u4v){a.x, a.y, b.x, b.y};
        t.r1 = (u4v){c.x, c.y, e.x, e.y};
    } else {
        // padded texel (x0 + 1, y0 + 1): (y0 * pitch + x0) + (pitch + 1) >= 0
        const uint32_t o = ((uint32_t)(__mul24(y0, (int)d.pitch) + x0) + (d.pitch + 1u)) << 3;
        t.r0 = gload_u4u(d.texels, o);
        t.r1 = gload_u4u(d.texels + (size_t)d.pitch * 8u, o);
    }
    tap_weights(fx, fy, t);
}
__device__ __forceinline__ void fetch_taps_plain(const TexS &d, float u, float v, Taps &t, bool q8 = false) {
    int x0, x1, y0, y1;
    float fx, fy;
    wrap_axis(u, d.w, x0, x1, fx, q8);
    wrap_axis(v, d.h, y0, y1, fy, q8);
    const uint32_t r0 = (uint32_t)y0 * d.w, r1 = (uint32_t)y1 * d.w;
    t.r0.x = gload_u32(d.texels, (r0 + (uint32_t)x0) * 4u); t.r0.z = gload_u32(d.texels, (r0 + (uint32_t)x1) * 4u);
    t.r1.x = gload_u32(d.texels, (r1 + (uint32_t)x0) * 4u); t.r1.z = gload_u32(d.texels, (r1 + (uint32_t)x1) * 4u);
    t.r0.y = t.r0.w = t.r1.y = t.r1.w = 0u;
    tap_weights(fx, fy, t);
}
// byte k of a word as float: v_cvt_f32_ubyteK, one instruction
template <int K> __device__ __forceinline__ float ubyte(uint32_t w) { return (float)((w >> (8 * K)) & 0xFFu); }
// UNORM8 channel (byte K of word W of the texel), bilinear, scaled to [0,1]
template <int W, int K> __device__ __forceinline__ float filt_bytes(const Taps &t) {   // the filtered channel on the 0..255 scale
    const uint32_t a = W ? t.r0.y : t.r0.x, b = W ? t.r0.w : t.r0.z, c = W ? t.r1.y : t.r1.x, d = W ? t.r1.w : t.r1.z;
    return fm(t.w11, ubyte<K>(d), fm(t.w01, ubyte<K>(c), fm(t.w10, ubyte<K>(b), t.w00 * ubyte<K>(a))));
}
template <int W, int K> __device__ __forceinline__ float filt_unorm(const Taps &t) { return filt_bytes<W, K>(t) * (1.0f / 255.0f); }
// 2 s / 255 - 1 for a normal-map channel s on the 0..255 scale, with 2/255 carried as two floats: the fp32 constant alone is 5.9e-8
// too large, a bias that tilts every normal by ~1e-7 towards its tangent-space offset -- nothing, except where n.wo is 1e-6 itself
// (grazing views: the specular term there is proportional to n.wo; measured +0.6 % of a pixel's radiance against the oracle)
__device__ __forceinline__ float snorm_of_bytes(float s) {
    constexpr float HI = 0x1.010102p-7f, LO = -0x1.fdfdfep-32f;   // HI + LO = 2/255 to 2^-56
    return fm(s, LO, fm(s, HI, -1.0f));
}
// sRGB8 channel (byte K of word 0), decoded per texel BEFORE filtering through the 256-entry LDS table
template <int K> __device__ __forceinline__ float filt_srgb(const Taps &t, const float *lut) {
    return fm(t.w11, lut[(t.r1.z >> (8 * K)) & 0xFFu], fm(t.w01, lut[(t.r1.x >> (8 * K)) & 0xFFu],
              fm(t.w10, lut[(t.r0.z >> (8 * K)) & 0xFFu], t.w00 * lut[(t.r0.x >> (8 * K)) & 0xFFu])));
}

// ---- forward.hlsl:68-96 calculate_shadow: 5x5 taps, each a bilinear fetch of the R32 map, WRAP -----
// Bit-exact against the oracle: the result is k/25 and one flipped comparison is a visible error.
__device__ __forceinline__ float lerp_exact(float a, float b, float t) { return __builtin_fmaf(t, b - a, a); }

__device__ __noinline__ float shadow_generic(const float *__restrict__ map, uint32_t S, float px, float py, float pz, bool q8 = false) {
#pragma clang fp contract(off)
    float shadow = 0.0f;
    for (int i = -2; i <= 2; ++i) {
        int x0, x1; float fx;
        wrap_axis(px + (float)i * 0.0001f, S, x0, x1, fx, q8);
        for (int j = -2; j <= 2; ++j) {
            int y0, y1; float fy;
            wrap_axis(py + (float)j * 0.0001f, S, y0, y1, fy, q8);
            const float *r0 = map + (size_t)y0 * S, *r1 = map + (size_t)y1 * S;
            float top = lerp_exact(r0[x0], r0[x1], fx), bot = lerp_exact(r1[x0], r1[x1], fx);
            float closest = lerp_exact(top, bot, fy);
            shadow += pz > closest ? 1.0f : 0.0f;
        }
    }
    return shadow / 25.0f;
}

// The 25 taps are 1e-4 apart in uv (0.4 texel at S = 4000), so with their bilinear neighbours they touch at most a 4x4
// texel window whenever S <= 5000.  Every bilinear result lies in [min, max] of the texels it reads (an fmaf lerp with a
// weight in [0,1) cannot leave the interval of its operands), so
//     pz > max  =>  all 25 taps shadowed,        pz <= min  =>  none
// decided without filtering, exactly.  Two levels of that test:
//   shadow_quick   bounds from the precomputed table (k_shadow_bounds): entry (i, j) = min/max over texels [4i, 4i+8) x
//                  [4j, 4j+8), which contains the 4x4 window of every footprint whose first texel is in [4i, 4i+4)^2:
//                  ONE 8-byte load per pixel, tap 0 is the only coordinate needed;
//   shadow_window  (tiles on a shadow edge) loads the window itself (4 x 16 B), tests its own min/max, and otherwise
//                  evaluates the 25 bilinear compares from registers in the oracle's operation order (horizontal lerps are
//                  shared between taps, which does not change any tap's value).
// Lanes near the map border (WRAP would engage) or with a wider footprint use shadow_generic.

// Q8: the D3D-style sampler for the taps (ARCTIC_OPT_SAMPLER bit 2; the oracle's SAMPLER_Q8_SHADOW): every scaled tap coordinate snapped to 1/256
// texel before it is split into texel and weight.  The window is then the one of the SNAPPED coordinates (a coordinate may snap across an
// integer); its span is checked like the plain one's, and whatever does not fit takes shadow_generic with the same sampler.
template <bool CANDIDATES = false, bool Q8 = false>
__device__ __forceinline__ float shadow_window(const float *__restrict__ map, uint32_t S, float px, float py, float pz, bool *tapped = nullptr /* statistics: this lane ran the 25 compares */) {
#pragma clang fp contract(off)
    const float Sf = (float)S;
    const auto scaled = [&](float u) { const float x = u * Sf - 0.5f; return Q8 ? snap256(x) : x; };
    // inside [0,1): u - floor(u) == u, so the coordinates below are wrap_axis without the wrap
    const float u0 = px + -0.0002f, u4 = px + 0.0002f, v0 = py + -0.0002f, v4 = py + 0.0002f;
    const float xa = floorf(scaled(u0)), xb = floorf(scaled(u4)), ya = floorf(scaled(v0)), yb = floorf(scaled(v4));
    const bool ok = u0 >= 0.0f && u4 < 1.0f && v0 >= 0.0f && v4 < 1.0f && xa >= 0.0f && ya >= 0.0f && xb - xa <= 2.0f && yb - ya <= 2.0f &&
                    xa + 3.0f < Sf && ya + 3.0f < Sf;
    if (!ok) return shadow_generic(map, S, px, py, pz, Q8);
    const uint32_t o0 = ((uint32_t)(int)ya * S + (uint32_t)(int)xa) * 4u;   // byte offset: maps are at most 16384^2 floats
    const float4u w0 = gload_f4u(map, o0), w1 = gload_f4u(map, o0 + S * 4u), w2 = gload_f4u(map, o0 + S * 8u), w3 = gload_f4u(map, o0 + S * 12u);
    const float lo = fminf(fminf(fminf(fminf(w0.x, w0.y), fminf(w0.z, w0.w)), fminf(fminf(w1.x, w1.y), fminf(w1.z, w1.w))),
                           fminf(fminf(fminf(w2.x, w2.y), fminf(w2.z, w2.w)), fminf(fminf(w3.x, w3.y), fminf(w3.z, w3.w))));
    const float hi = fmaxf(fmaxf(fmaxf(fmaxf(w0.x, w0.y), fmaxf(w0.z, w0.w)), fmaxf(fmaxf(w1.x, w1.y), fmaxf(w1.z, w1.w))),
                           fmaxf(fmaxf(fmaxf(w2.x, w2.y), fmaxf(w2.z, w2.w)), fmaxf(fmaxf(w3.x, w3.y), fmaxf(w3.z, w3.w))));
    if (pz > hi) return 1.0f;
    if (!(pz > lo)) return 0.0f;
    if (tapped) *tapped = true;
    // The 25 compares, one tap column at a time.  A tap's texels are whichever CELL of the window it falls into, per lane -- selecting the two
    // operands of every lerp costs four v_cndmask (4.3 issue cycles each) per lerp.  Round 5: each lerp is evaluated in all three candidate
    // cells instead -- fmaf(t, b - a, a) with the differences b - a formed once per row (the oracle's own subtraction), three fused
    // multiply-adds at 2.9 cycles -- and the RESULT of the lane's cell is selected (two v_cndmask): the same operations on the same operands
    // for the cell that counts, so the same bits; 17 issue cycles per lerp instead of 23.
    float fy[5];
    bool r0[5], r1[5];   // row of tap j relative to the window: 0 / 1 / 2
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const float y = scaled(py + (float)(j - 2) * 0.0001f), yf = floorf(y);
        fy[j] = y - yf;
        r0[j] = yf == ya; r1[j] = yf == ya + 1.0f;
    }
    float shadow = 0.0f;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const float x = scaled(px + (float)(i - 2) * 0.0001f), xf = floorf(x), fx = x - xf;
        const bool c0 = xf == xa, c1 = xf == xa + 1.0f;
        const auto pick = [](bool p0, bool p1, float v0, float v1, float v2) { return p0 ? v0 : (p1 ? v1 : v2); };
        float h0, h1, h2, h3;
        if (CANDIDATES && ARCTIC_PCF_ROW_CANDIDATES) {
            // (the horizontal lerps the same way keep the window's twelve differences alive through all five columns: 76 VGPRs, a wave per SIMD lost)
            h0 = pick(c0, c1, __builtin_fmaf(fx, w0.y - w0.x, w0.x), __builtin_fmaf(fx, w0.z - w0.y, w0.y), __builtin_fmaf(fx, w0.w - w0.z, w0.z));
            h1 = pick(c0, c1, __builtin_fmaf(fx, w1.y - w1.x, w1.x), __builtin_fmaf(fx, w1.z - w1.y, w1.y), __builtin_fmaf(fx, w1.w - w1.z, w1.z));
            h2 = pick(c0, c1, __builtin_fmaf(fx, w2.y - w2.x, w2.x), __builtin_fmaf(fx, w2.z - w2.y, w2.y), __builtin_fmaf(fx, w2.w - w2.z, w2.z));
            h3 = pick(c0, c1, __builtin_fmaf(fx, w3.y - w3.x, w3.x), __builtin_fmaf(fx, w3.z - w3.y, w3.y), __builtin_fmaf(fx, w3.w - w3.z, w3.z));
        } else {
            h0 = lerp_exact(pick(c0, c1, w0.x, w0.y, w0.z), pick(c0, c1, w0.y, w0.z, w0.w), fx);
            h1 = lerp_exact(pick(c0, c1, w1.x, w1.y, w1.z), pick(c0, c1, w1.y, w1.z, w1.w), fx);
            h2 = lerp_exact(pick(c0, c1, w2.x, w2.y, w2.z), pick(c0, c1, w2.y, w2.z, w2.w), fx);
            h3 = lerp_exact(pick(c0, c1, w3.x, w3.y, w3.z), pick(c0, c1, w3.y, w3.z, w3.w), fx);
        }
        if (CANDIDATES) {
            const float e0 = h1 - h0, e1 = h2 - h1, e2 = h3 - h2;   // b - a down the column, for the three cells
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const float closest = pick(r0[j], r1[j], __builtin_fmaf(fy[j], e0, h0), __builtin_fmaf(fy[j], e1, h1), __builtin_fmaf(fy[j], e2, h2));
                shadow += pz > closest ? 1.0f : 0.0f;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const float closest = lerp_exact(pick(r0[j], r1[j], h0, h1, h2), pick(r0[j], r1[j], h1, h2, h3), fy[j]);
                shadow += pz > closest ? 1.0f : 0.0f;
            }
        }
    }
    return shadow / 25.0f;
}

// (Round 4 also tried the lane's 4x4 window in LDS -- texel k of lane l at word 64 k + l, a tap = two ds_read2st64_b32 at a computed address, ~290 vector
// instructions + 58 LDS operations instead of ~425 + 0 -- and measured it slower: profiles/r4_d_ab_window_in_lds.txt; code in commit ec02d97.)

// ---- the LDS variant of the 25-tap path (north_star: "stages ... shadow-map tiles in LDS"; ARCTIC_OPT_DEBUG bit 4 selects the
// kernels instantiated with it) ----------------------------------------------------------------------------------------------
// For a tile on a shadow edge: the bounding box of the undecided lanes' footprints (4x4 windows) is staged ONCE per wave in LDS
// (<= 32 x 32 texels; the lanes of an 8x8 screen tile land within a few texels of each other) and every tap reads its four texels
// from there with computed addresses -- no selects: per tap 2 ds_read2_b32 + 3 exact lerps + compare, in the oracle's order.
// Against the register window (4 x 16-byte loads per lane, ~185 v_cndmask to pick texels): measured in DESIGN.md section 4.2.
constexpr int SHADOW_TILE = 32;   // texels per side of a wave's LDS tile
__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = min(v, __shfl_xor(v, d));
    return v;
}
// returns true when the wave's undecided lanes were handled here (lit updated); false: the caller takes the register path
__device__ __forceinline__ bool shadow_lds_tile(const ShadowArgs &sa, float *tile /* this wave's SHADOW_TILE^2 floats */, uint32_t lane, bool undecided,
                                                const float px, const float py, const float pz, float &lit) {
#pragma clang fp contract(off)
    const uint32_t S = sa.S;
    const float Sf = (float)S;
    const int bx = floor_to_int((px + -0.0002f) * Sf - 0.5f), by = floor_to_int((py + -0.0002f) * Sf - 0.5f);
    const int ex = floor_to_int((px + 0.0002f) * Sf - 0.5f) + 1, ey = floor_to_int((py + 0.0002f) * Sf - 0.5f) + 1;   // last texel a tap reads
    const bool ok = bx >= 0 && by >= 0 && ex < (int)S && ey < (int)S;   // no tap wraps (the window test of shadow_window, without its width limit)
    if (__ballot(undecided && !ok) != 0ull) return false;
    const int BIG = 1 << 30;
    const int x0 = wave_min_i32(undecided ? bx : BIG), y0 = wave_min_i32(undecided ? by : BIG);
    const int x1 = -wave_min_i32(undecided ? -ex : BIG), y1 = -wave_min_i32(undecided ? -ey : BIG);
    const int W = x1 - x0 + 1, H = y1 - y0 + 1;
    if (W > SHADOW_TILE || H > SHADOW_TILE) return false;
    // stage rows y0 .. y1, texels x0 .. x1: two rows per step (lanes 0-31 / 32-63), coalesced along x
    for (int r = (int)(lane >> 5); r < H; r += 2) {
        const int c = (int)(lane & 31);
        if (c < W) tile[r * SHADOW_TILE + c] = sa.map[(size_t)(y0 + r) * S + (size_t)(x0 + c)];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // same wave: LDS operations complete in order
    if (undecided) {
        float shadow = 0.0f;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const float x = (px + (float)(i - 2) * 0.0001f) * Sf - 0.5f, xf = floorf(x), fx = x - xf;
            const int cx = (int)xf - x0;
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const float y = (py + (float)(j - 2) * 0.0001f) * Sf - 0.5f, yf = floorf(y), fy = y - yf;
                const float *t0 = tile + ((int)yf - y0) * SHADOW_TILE + cx;
                const float top = lerp_exact(t0[0], t0[1], fx), bot = lerp_exact(t0[SHADOW_TILE], t0[SHADOW_TILE + 1], fx);
                shadow += pz > lerp_exact(top, bot, fy) ? 1.0f : 0.0f;
            }
        }
        lit = 1.0f - shadow / 25.0f;
    }
    return true;
}

// 1 - shadow in two steps.  shadow_quick decides from the bounds table where it can (and for every pixel outside the map);
// returns false for the lanes that need shadow_slow: tiles on a shadow edge, the map's border, maps above 5000^2.
// (ShadowPos, shadow_coords, shadow_table_offset: shadow_coords.h, shared with the prepass's cost hint)
__device__ __forceinline__ bool shadow_table_entry(const ShadowArgs &sa, const ShadowPos &p, uint32_t &offset, bool q8) { return shadow_table_offset(sa.S, sa.pitch, p, offset, q8 ? 4u : 3u); }
// in two steps, so that a caller can put other work (a batch of scalar loads) between the table load and its use:
//   shadow_quick_issue   coordinates + the load of the table entry; returns whether the pixel is within the table's reach
//   shadow_quick_decide  lit = 0 / 1 where the entry (or the map's border rule) decides; false: the pixel needs shadow_slow
__device__ __forceinline__ bool shadow_quick_issue(const ShadowArgs &sa, float lsx, float lsy, float lsz, float lsw, ShadowPos &p, float2 &mm, uint32_t *offset_out = nullptr, bool q8 = false /* wave-uniform */) {
    mm = make_float2(0.0f, 0.0f);
    if (sa.map == nullptr) return false;
    shadow_coords(lsx, lsy, lsz, lsw, p);
    uint32_t offset = 0;
    const bool in_table = sa.bounds != nullptr && shadow_table_entry(sa, p, offset, q8);   // (a table only for S <= 4900: shadow_bounds_pitch)
    if (in_table) mm = gload_f2(sa.bounds, offset);
    if (offset_out) *offset_out = offset;
    return in_table;
}
__device__ __forceinline__ bool shadow_quick_decide(const ShadowArgs &sa, bool in_table, const ShadowPos &p, float2 mm, float &lit) {
    lit = 1.0f;
    if (sa.map == nullptr) return true;
    if (in_table) {
        if (p.pz > mm.y) { lit = 0.0f; return true; }
        return !(p.pz > mm.x);
    }
    // outside the map: no shadow (forward.hlsl:75-77); everything else takes the slow path
    return p.pz > 1.0f || p.px < 0.0f || p.py < 0.0f || p.px > 1.0f || p.py > 1.0f;
}
__device__ __forceinline__ bool shadow_quick(const ShadowArgs &sa, float lsx, float lsy, float lsz, float lsw, ShadowPos &p, float &lit, bool q8 = false) {
    float2 mm;
    const bool in_table = shadow_quick_issue(sa, lsx, lsy, lsz, lsw, p, mm, nullptr, q8);
    return shadow_quick_decide(sa, in_table, p, mm, lit);
}
__device__ __forceinline__ float shadow_slow(const ShadowArgs &sa, const ShadowPos &p, bool q8 = false /* wave-uniform */) {
    const uint32_t S = sa.S;
    if (q8) { asm volatile(""); return 1.0f - (S <= 5000u ? shadow_window<false, true>(sa.map, S, p.px, p.py, p.pz) : shadow_generic(sa.map, S, p.px, p.py, p.pz, true)); }
    return 1.0f - (S <= 5000u ? shadow_window(sa.map, S, p.px, p.py, p.pz) : shadow_generic(sa.map, S, p.px, p.py, p.pz));
}

// ---- forward.hlsl:126-193 -----------------------------------------------------------------------
// calculate_outgoing_radiance with everything that does not depend on the light hoisted out of the loop.
//
// Per light only three scalars feed the colour channels: with F = F0 + (1 - F0) p5 (Schlick, :126-129) the term
//     (kD base / PI + spec) Li n.wi,   kD = (1 - F)(1 - metal)
// is   kdb (1 - F0) [colour s1] + num F0 [colour s2] + num (1 - F0) [colour s3]    with
//     s1 = sc (1 - p5),  s2 = sc g,  s3 = s2 p5,   sc = n.wi / |d|^2,   g = spec / num,
// kdb = (1 - metal) base / PI and num = a2 g1(n.wo) / PI per pixel.  The loops accumulate colour x {s1, s2, s3} (9 sums) and
// the per-pixel factors are applied once (resolve_sums).
//
// Conditioning: the GGX denominator n_dot_h^2 (a2 - 1) + 1 (:137) cancels to ~a2 at a highlight; written as
// sin^2 (1 - a2) + a2 it has no cancellation (same value in exact arithmetic), with sin^2 from e = n - h/|h|:
// |e|^2 = 2 - 2 cos, sin^2 = |e|^2 (1 - |e|^2 / 4).  h = wo + wi is formed component-wise like the HLSL.
// (h . wo) / |h| = |h| / 2 for unit wo, wi, so Schlick's 1 - h.wo = 1 - |h| / 2, in [0,1] by construction.
// G's and the BRDF's denominators (x (1 - k) + k)(4 n.wo x + 1e-4), x = n.wi, are one quadratic q2 x^2 + q1 x + q0 (:153,:172).
struct LoopPix {
    f3 n, wo, world;
    float a2, oma2, c4;   // roughness^4, 1 - roughness^4, -(1 - roughness^4) / 4
    float q2, q1, q0;
};
// What the tail needs of a pixel is kept through the light loop as FIVE values (base colour, metalness, num) and turned into the
// per-channel factors F0, 1 - F0, kdb only behind the loop (tail_factors): nine registers fewer across it than the factors themselves
// (round 4: with them the kernel spilled 8 bytes per lane around the loop, i.e. needed a scratch allocation for every wave).
struct TailPix {
    f3 base;
    float metal, num;
};
struct TailFactors { f3 F0, omF0, kdb; float num; };
__device__ __forceinline__ TailFactors tail_factors(const TailPix &t) {
    TailFactors f;
    f.F0 = mk(fm(t.metal, t.base.x - 0.04f, 0.04f), fm(t.metal, t.base.y - 0.04f, 0.04f), fm(t.metal, t.base.z - 0.04f, 0.04f));   // lerp(0.04, base, metal) :181-182
    f.omF0 = mk(1.0f - f.F0.x, 1.0f - f.F0.y, 1.0f - f.F0.z);
    const float km = (1.0f - t.metal) * INV_PI;
    f.kdb = mk(t.base.x * km, t.base.y * km, t.base.z * km);
    f.num = t.num;
    return f;
}
__device__ __forceinline__ void make_pix(f3 n, f3 wo, f3 world, f3 base, float metal, float rough, LoopPix &p, TailPix &t) {
    p.n = n; p.wo = wo; p.world = world;
    t.base = base; t.metal = metal;
    const float ndwo = fmaxf(dot(n, wo), 0.0f);
    const float a = rough * rough, a2 = a * a;
    p.a2 = a2; p.oma2 = 1.0f - a2; p.c4 = -0.25f * p.oma2;
    const float r1 = rough + 1.0f;
    const float k = r1 * r1 * 0.125f, omk = 1.0f - k;
    t.num = a2 * INV_PI * ndwo * rcp(fm(ndwo, omk, k));
    asm volatile("" : "+v"(t.num));   // one register through the light loop instead of the four it is made of (the compiler would sink the division behind the loop)
    const float four_ndwo = 4.0f * ndwo;
    p.q2 = omk * four_ndwo; p.q1 = fm(k, four_ndwo, 0.0001f * omk); p.q0 = 0.0001f * k;
}

// the three per-light scalars for light direction d (unnormalised, towards the light), nd = n.d.  POINT: radiance = colour /
// |d|^2 (forward.hlsl:226-229); otherwise d is unit and there is no falloff.
template <bool POINT>
__device__ __forceinline__ void light_scalars(const LoopPix &p, f3 d, float nd, float &s1, float &s2, float &s3) {
    float inv = 1.0f, inv2 = 1.0f;
    if (POINT) { inv = rsq(dot(d, d)); inv2 = inv * inv; }
    const float ndwi = sat(nd * inv);          // max(n . wi, 0); n and wi are unit vectors: the upper clamp only catches rounding
    const f3 h = mk(__builtin_fmaf(d.x, inv, p.wo.x), __builtin_fmaf(d.y, inv, p.wo.y), __builtin_fmaf(d.z, inv, p.wo.z));
    const float hh = dot(h, h), rh = rsq(hh);          // |h|^2, 1 / |h|
    const float m = __builtin_fmaf(hh * rh, -0.5f, 1.0f);
    const float m2 = m * m, p5 = m2 * m2 * m;
    const f3 e = mk(__builtin_fmaf(h.x, -rh, p.n.x), __builtin_fmaf(h.y, -rh, p.n.y), __builtin_fmaf(h.z, -rh, p.n.z));   // n - h/|h|
    const float e2 = dot(e, e);
    // (when n.h <= 0 the HLSL's max(n.h, 0) makes the denominator 1, but then n.wo <= 0 or n.wi <= 0 and the term is multiplied by 0 anyway)
    const float dd = __builtin_fmaf(e2, __builtin_fmaf(e2, p.c4, p.oma2), p.a2);   // sin^2 (1 - a2) + a2
    const float den = (dd * dd) * __builtin_fmaf(__builtin_fmaf(p.q2, ndwi, p.q1), ndwi, p.q0);
    const float sc = ndwi * inv2;
    s2 = (sc * ndwi) * rcp(den);
    s1 = __builtin_fmaf(-sc, p5, sc);
    s3 = s2 * p5;
}

typedef float v2 __attribute__((ext_vector_type(2)));

// ---- the packed light loop: two point lights per trip, {light a, light b} in the halves of every 64-bit operand ----------
// v_pk_{add,mul,fma}_f32 read 64-bit register pairs and pick, per operand, which half feeds the low and the high result
// (op_sel / op_sel_hi).  Per-pixel values are the same for both lights: instead of a duplicated {x, x} pair per value
// (15 values = 30 registers, what the compiler makes of a splat) two DIFFERENT per-pixel values share one pair and the
// instruction broadcasts the half it needs -- 16 registers, which is what lets the kernel keep 6 waves per SIMD.  The light
// pairs are wave-uniform and come through the scalar cache into SGPR pairs (one SGPR source per instruction is allowed),
// double buffered, so the loop has no LDS or vector-memory instruction at all.
// Naming: _b = operand broadcast from half H of its pair.
#define PK_SEL(H) template <> __device__ __forceinline__
template <int H> __device__ __forceinline__ v2 pk_sub_sb(v2 l, v2 w);                // l (SGPR pair) - w[H]
template <int H> __device__ __forceinline__ v2 pk_mul_b(v2 a, v2 b);                 // a * b[H]
template <int H> __device__ __forceinline__ v2 pk_fma_b1(v2 a, v2 b, v2 c);          // a * b[H] + c
template <int H> __device__ __forceinline__ v2 pk_fma_b2(v2 a, v2 b, v2 c);          // a * b + c[H]
template <int H> __device__ __forceinline__ v2 pk_fnma_b2(v2 a, v2 b, v2 c);         // -a * b + c[H]
template <int H1, int H2> __device__ __forceinline__ v2 pk_fma_b12(v2 a, v2 b, v2 c);   // a * b[H1] + c[H2]
#define PK_DEFS(H)                                                                                                                          \
    PK_SEL(H) v2 pk_sub_sb<H>(v2 l, v2 w) { v2 r; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0," #H "] op_sel_hi:[1," #H "] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "s"(l), "v"(w)); return r; } \
    PK_SEL(H) v2 pk_mul_b<H>(v2 a, v2 b) { v2 r; asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0," #H "] op_sel_hi:[1," #H "]" : "=v"(r) : "v"(a), "v"(b)); return r; } \
    PK_SEL(H) v2 pk_fma_b1<H>(v2 a, v2 b, v2 c) { v2 r; asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0," #H ",0] op_sel_hi:[1," #H ",1]" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; } \
    PK_SEL(H) v2 pk_fma_b2<H>(v2 a, v2 b, v2 c) { v2 r; asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0," #H "] op_sel_hi:[1,1," #H "]" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; } \
    PK_SEL(H) v2 pk_fnma_b2<H>(v2 a, v2 b, v2 c) { v2 r; asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0," #H "] op_sel_hi:[1,1," #H "] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
PK_DEFS(0)
PK_DEFS(1)
#define PK_DEFS2(H1, H2)                                                                                                                    \
    template <> __device__ __forceinline__ v2 pk_fma_b12<H1, H2>(v2 a, v2 b, v2 c) { v2 r; asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0," #H1 "," #H2 "] op_sel_hi:[1," #H1 "," #H2 "]" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
PK_DEFS2(0, 0)
PK_DEFS2(0, 1)
PK_DEFS2(1, 0)
PK_DEFS2(1, 1)
__device__ __forceinline__ v2 pk_fma(v2 a, v2 b, v2 c) { return __builtin_elementwise_fma(a, b, c); }
// acc += colour (SGPR pair) * s, IN PLACE: with a separate result register the compiler renames the 18 sums between the two halves of the
// unrolled loop -- 18 copies at one of its exits and a second set of 18 zeroes in front of it (round 5: read off the ISA)
__device__ __forceinline__ void pk_fma_s(v2 c, v2 s, v2 &acc) { asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "s"(c), "v"(s)); }
// v_pk_mul_f32 with the clamp bit: sat(a * b) for both halves in one instruction (no packed max/min exists for fp32)
__device__ __forceinline__ v2 pk_mul_sat(v2 a, v2 b) { v2 r; asm("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b)); return r; }
// Transcendentals.  gfx950 needs one wait state between a transcendental and a VALU instruction that reads its result; the
// compiler inserts it for instructions it generates but does not look inside asm statements, and every consumer here is
// one.  So each pair of transcendentals is ONE asm statement that ends with an independent packed operation of the same
// loop body (work that has to be done anyway) -- whatever comes next is at least one instruction away from the result.
__device__ __forceinline__ v2 rsq2_and_mul_b0(v2 a, v2 x, v2 y, v2 &t) {            // rsq(a) | t = x * y[0]
    float r0, r1;
    asm("v_rsq_f32 %0, %3\n\tv_rsq_f32 %1, %4\n\tv_pk_mul_f32 %2, %5, %6 op_sel:[0,0] op_sel_hi:[1,0]"
        : "=&v"(r0), "=&v"(r1), "=&v"(t) : "v"(a.x), "v"(a.y), "v"(x), "v"(y));
    return (v2){r0, r1};
}
__device__ __forceinline__ v2 rsq2_and_fma_b01(v2 a, v2 x, v2 y, v2 &t) {           // rsq(a) | t = x * y[0] + y[1]
    float r0, r1;
    asm("v_rsq_f32 %0, %3\n\tv_rsq_f32 %1, %4\n\tv_pk_fma_f32 %2, %5, %6, %6 op_sel:[0,0,1] op_sel_hi:[1,0,1]"
        : "=&v"(r0), "=&v"(r1), "=&v"(t) : "v"(a.x), "v"(a.y), "v"(x), "v"(y));
    return (v2){r0, r1};
}
__device__ __forceinline__ v2 rcp2_and_mul(v2 a, v2 x, v2 y, v2 &t) {               // rcp(a) | t = x * y
    float r0, r1;
    asm("v_rcp_f32 %0, %3\n\tv_rcp_f32 %1, %4\n\tv_pk_mul_f32 %2, %5, %6"
        : "=&v"(r0), "=&v"(r1), "=&v"(t) : "v"(a.x), "v"(a.y), "v"(x), "v"(y));
    return (v2){r0, r1};
}

struct Sums { float a[3], b[3], c[3]; };                     // scalar loop
struct Sums2 { v2 a[3], b[3], c[3]; };                       // packed loop: .x/.y = the two lights of a pair
__device__ __forceinline__ f3 resolve_sums(const TailPix &tp, const float A[3], const float B[3], const float C[3]) {
    const TailFactors t = tail_factors(tp);
    return mk(fm(t.kdb.x * t.omF0.x, A[0], t.num * fm(t.F0.x, B[0], t.omF0.x * C[0])),
              fm(t.kdb.y * t.omF0.y, A[1], t.num * fm(t.F0.y, B[1], t.omF0.y * C[1])),
              fm(t.kdb.z * t.omF0.z, A[2], t.num * fm(t.F0.z, B[2], t.omF0.z * C[2])));
}

// the per-pixel constants of the packed loop, two to a register pair
struct PackedPix { v2 n_xy, nz_wox, wo_yz, w_xy, wz_a2, oma2_c4, q21, q0_; };
__device__ __forceinline__ PackedPix pack_pix(const LoopPix &p) {
    PackedPix k;
    k.n_xy = (v2){p.n.x, p.n.y}; k.nz_wox = (v2){p.n.z, p.wo.x}; k.wo_yz = (v2){p.wo.y, p.wo.z};
    k.w_xy = (v2){p.world.x, p.world.y}; k.wz_a2 = (v2){p.world.z, p.a2}; k.oma2_c4 = (v2){p.oma2, p.c4};
    k.q21 = (v2){p.q2, p.q1}; k.q0_ = (v2){p.q0, 0.0f};
    return k;
}
// scalar loads of one light pair, and the wait that makes them usable (tied to the registers, so every use comes after it)
typedef float f4v __attribute__((ext_vector_type(4)));
// (volatile: the two statements keep their order, so the wait never covers loads issued after it in program order)
__device__ __forceinline__ void load_light_pair(const char *p /* wave-uniform */, f4v &A, f4v &B, f4v &C) {
    asm volatile("s_load_dwordx4 %0, %3, 0x0\n\ts_load_dwordx4 %1, %3, 0x10\n\ts_load_dwordx4 %2, %3, 0x20" : "=&s"(A), "=&s"(B), "=&s"(C) : "s"(p));
}
// d = l - world for the three components of a light pair, behind the s_waitcnt that makes the pair's SGPRs valid: every
// other use of the pair depends on d, so it is ordered after the wait by data flow
__device__ __forceinline__ void wait_and_sub(v2 lx, v2 ly, v2 lz, v2 w_xy, v2 wz, v2 &dx, v2 &dy, v2 &dz) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\t"
                 "v_pk_add_f32 %0, %3, %6 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                 "v_pk_add_f32 %1, %4, %6 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                 "v_pk_add_f32 %2, %5, %7 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]"
                 : "=&v"(dx), "=&v"(dy), "=&v"(dz) : "s"(lx), "s"(ly), "s"(lz), "v"(w_xy), "v"(wz));
}

// light_scalars<true> for the two point lights of a pair (d = position - world from wait_and_sub, colours cr, cg, cb in SGPR
// pairs): lane-wise the same arithmetic as the scalar loop.  nd (n . d of both lights) is returned for the statistics.
__device__ __forceinline__ v2 accumulate_pair(const PackedPix &k, v2 dx, v2 dy, v2 dz, v2 cr, v2 cg, v2 cb, Sums2 &S) {
    const v2 d2 = pk_fma(dz, dz, pk_fma(dy, dy, dx * dx));
    v2 ndx;                                                                     // n.x d.x, in the shadow of the rsq
    const v2 inv = rsq2_and_mul_b0(d2, dx, k.n_xy, ndx);
    const v2 nd = pk_fma_b1<0>(dz, k.nz_wox, pk_fma_b1<1>(dy, k.n_xy, ndx));
    const v2 ndwi = pk_mul_sat(nd, inv);
    const v2 hx = pk_fma_b2<1>(dx, inv, k.nz_wox), hy = pk_fma_b2<0>(dy, inv, k.wo_yz), hz = pk_fma_b2<1>(dz, inv, k.wo_yz);
    const v2 hh = pk_fma(hz, hz, pk_fma(hy, hy, hx * hx));
    v2 q;                                                                       // q2 x + q1, x = n.wi
    const v2 rh = rsq2_and_fma_b01(hh, ndwi, k.q21, q);
    const v2 m = pk_fma(hh * rh, (v2){-0.5f, -0.5f}, (v2){1.0f, 1.0f});
    const v2 ex = pk_fnma_b2<0>(hx, rh, k.n_xy), ey = pk_fnma_b2<1>(hy, rh, k.n_xy), ez = pk_fnma_b2<0>(hz, rh, k.nz_wox);   // n - h/|h|
    const v2 e2 = pk_fma(ez, ez, pk_fma(ey, ey, ex * ex));
    const v2 dd = pk_fma_b2<1>(e2, pk_fma_b12<1, 0>(e2, k.oma2_c4, k.oma2_c4), k.wz_a2);   // a2 + e2 (oma2 + c4 e2)
    const v2 den = (dd * dd) * pk_fma_b2<0>(q, ndwi, k.q0_);                               // dd^2 ((q2 x + q1) x + q0)
    const v2 sc = ndwi * (inv * inv);
    v2 m2;
    const v2 rden = rcp2_and_mul(den, m, m, m2);
    const v2 p5 = m2 * m2 * m;
    const v2 s2 = (sc * ndwi) * rden;
    const v2 s1 = pk_fma(-sc, p5, sc), s3 = s2 * p5;
    pk_fma_s(cr, s1, S.a[0]); pk_fma_s(cg, s1, S.a[1]); pk_fma_s(cb, s1, S.a[2]);
    pk_fma_s(cr, s2, S.b[0]); pk_fma_s(cg, s2, S.b[1]); pk_fma_s(cb, s2, S.b[2]);
    pk_fma_s(cr, s3, S.c[0]); pk_fma_s(cg, s3, S.c[1]); pk_fma_s(cb, s3, S.c[2]);
    return nd;
}

// ---- post_process.hlsl ---------------------------------------------------------------------------
__device__ __forceinline__ float rrt_odt(float c) {
    float a = fm(c, c + 0.0245786f, -0.000090537f);
    float b = fm(c, fm(0.983729f, c, 0.4329510f), 0.238081f);
    return a * rcp(b);
}
// the last multiply-add of the ACES output matrix with the saturate (:24) in the instruction's clamp bit
__device__ __forceinline__ v2 pk_fma_sat(v2 k /* SGPR pair */, v2 a, v2 c) { v2 r; asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(r) : "s"(k), "v"(a), "v"(c)); return r; }
__device__ __forceinline__ float fma_sat(float k /* SGPR */, float a, float c) { float r; asm("v_fma_f32 %0, %1, %2, %3 clamp" : "=v"(r) : "s"(k), "v"(a), "v"(c)); return r; }
// the tonemappers, post_process.hlsl:39-57 (what correct_gamma is applied to)
__device__ __forceinline__ f3 tonemap(f3 c, int tm, float exposure) {
    f3 t;
    if (tm == 1) {          // tm_exposure :44-47: 1 - exp(-c * exposure); for tiny arguments the subtraction cancels in
                            // fp32 (1 - exp(-1e-8) = 0), so the series x - x^2/2 + x^3/6 takes over below 1/64
        const float LOG2E = 1.4426950408889634f;
        const f3 x = mk(c.x * exposure, c.y * exposure, c.z * exposure);
        auto one_minus_exp = [&](float v) {
            const float direct = 1.0f - __builtin_amdgcn_exp2f(-v * LOG2E);
            const float series = v * fm(v, fm(v, 1.0f / 6.0f, -0.5f), 1.0f);
            return fabsf(v) < 0.015625f ? series : direct;
        };
        t = mk(one_minus_exp(x.x), one_minus_exp(x.y), one_minus_exp(x.z));
    } else if (tm == 2) {   // tm_aces :15-25, :50-57; channels x and y as a packed pair (same operations lane for lane), z plain
        auto pfm = [](v2 a, v2 b, v2 c2) { return __builtin_elementwise_fma(a, b, c2); };
        const v2 cx = {c.x, c.x}, cy = {c.y, c.y}, cz = {c.z, c.z};
        v2 ixy = pfm((v2){0.04823f, 0.01566f}, cz, pfm((v2){0.35458f, 0.90834f}, cy, (v2){0.59719f, 0.07600f} * cx));
        float iz = fm(0.837f, c.z, fm(0.13383f, c.y, 0.02840f * c.x));
        {   // rrt_and_odt_fit
            const v2 a = pfm(ixy, ixy + (v2){0.0245786f, 0.0245786f}, (v2){-0.000090537f, -0.000090537f});
            const v2 b = pfm(ixy, pfm((v2){0.983729f, 0.983729f}, ixy, (v2){0.4329510f, 0.4329510f}), (v2){0.238081f, 0.238081f});
            ixy = a * (v2){rcp(b.x), rcp(b.y)};
            iz = rrt_odt(iz);
        }
        const v2 ix = {ixy.x, ixy.x}, iy = {ixy.y, ixy.y}, izz = {iz, iz};
        const v2 oxy = pk_fma_sat((v2){-0.07367f, -0.00605f}, izz, pfm((v2){-0.53108f, 1.10813f}, iy, (v2){1.60475f, -0.10208f} * ix));
        t = mk(oxy.x, oxy.y, fma_sat(1.07f, iz, fm(-0.07276f, ixy.y, -0.00327f * ixy.x)));
    } else {                // tm_reinhard :39-42 (and `default:`)
        t = mk(c.x * rcp(c.x + 1.0f), c.y * rcp(c.y + 1.0f), c.z * rcp(c.z + 1.0f));
    }
    return t;
}
// correct_gamma :34-37: pow(abs(t), 1/gamma) = exp2(g), g = log2|t| / gamma
__device__ __forceinline__ f3 gamma_exponent(f3 t, float inv_gamma) {
    return mk(inv_gamma * __builtin_amdgcn_logf(fabsf(t.x)), inv_gamma * __builtin_amdgcn_logf(fabsf(t.y)), inv_gamma * __builtin_amdgcn_logf(fabsf(t.z)));
}
__device__ __forceinline__ f3 post_process(f3 c, int tm, float inv_gamma, float exposure) {
    const f3 g = gamma_exponent(tonemap(c, tm, exposure), inv_gamma);
    return mk(__builtin_amdgcn_exp2f(g.x), __builtin_amdgcn_exp2f(g.y), __builtin_amdgcn_exp2f(g.z));
}
// store to the R8G8B8A8_UNORM target (renderer.cpp:161-175): saturate (NaN -> 0), *255, +0.5, truncate
// (v_med3_f32 of (NaN, 0, 1) is 0; v_cvt_u32_f32 truncates)
__device__ __forceinline__ uint32_t unorm8(float x) { return (uint32_t)__builtin_fmaf(sat(x), 255.0f, 0.5f); }
__device__ __forceinline__ uint32_t rgba8_word(f3 l) { return unorm8(l.x) | (unorm8(l.y) << 8) | (unorm8(l.z) << 16) | 0xFF000000u; }
// the same store for a colour given as its gamma exponents (LDR = exp2(g)): the saturate is the clamp bit of v_exp_f32, and the
// three multiply-adds follow the three transcendentals in ONE asm statement, so each reads its exponential two instructions
// after it was issued (gfx950 needs one wait state between a transcendental and the VALU instruction reading its result; the
// compiler does not look inside asm statements).  Bit for bit unorm8 of exp2(g): the clamp of a NaN is 0 (DX10_CLAMP).
__device__ __forceinline__ uint32_t rgba8_word_of_exponents(f3 g) {
    float e0, e1, e2, q0, q1, q2;
    asm("v_exp_f32_e64 %0, %6 clamp\n\tv_exp_f32_e64 %1, %7 clamp\n\tv_exp_f32_e64 %2, %8 clamp\n\t"
        "v_fma_f32 %3, %0, %9, 0.5\n\tv_fma_f32 %4, %1, %9, 0.5\n\tv_fma_f32 %5, %2, %9, 0.5"
        : "=&v"(e0), "=&v"(e1), "=&v"(e2), "=&v"(q0), "=&v"(q1), "=&v"(q2) : "v"(g.x), "v"(g.y), "v"(g.z), "s"(255.0f));
#if ARCTIC_CVT_PK_U8
    return __builtin_amdgcn_cvt_pk_u8_f32(q2, 2u, __builtin_amdgcn_cvt_pk_u8_f32(q1, 1u, __builtin_amdgcn_cvt_pk_u8_f32(q0, 0u, 0xFF000000u)));
#else
    return (uint32_t)q0 | ((uint32_t)q1 << 8) | ((uint32_t)q2 << 16) | 0xFF000000u;
#endif
}

// first wave of G-buffer loads: what every pixel needs (28 B)
struct TileHead { float4 a; float b0, b1, b2; };   // a = uv.xy, ls.xy; b = ls.z, ls.w, material id
// (the tile's base addresses are wave-uniform -- scalar 64-bit arithmetic -- and the lane adds a 32-bit offset: saddr loads)
typedef float f3v __attribute__((ext_vector_type(3), aligned(4)));
typedef float f4a __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 gload_f4(const void *base, uint32_t o) { const f4a v = *(const f4a __attribute__((address_space(1))) *)((gchar)base + o); return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ TileHead load_head(const float4 *ga, const float *gb, size_t tile /* wave-uniform */, uint32_t lane) {
    TileHead t;
    t.a = gload_f4(ga + tile * 64, lane * 16u);
    const f3v b = *(const f3v __attribute__((address_space(1))) *)((gchar)(gb + tile * 192) + lane * 12u);
    t.b0 = b.x; t.b1 = b.y; t.b2 = b.z;
    return t;
}

// the reference renders ps_main into an R16G16B16A16_FLOAT target (forward_pass.cpp:149) that post_process then reads:
// hdr16 reproduces that rounding (round-to-nearest-even to binary16, finite overflow to +inf like the ROP's conversion)
__device__ __forceinline__ float through_half(float x) { return (float)(_Float16)x; }

// post_process + the store of one pixel.  `out` = the RGBA8 target advanced to the tile's first pixel by the caller (wave-uniform,
// or the target itself), o = the pixel's index behind it (32-bit: targets are at most 16384^2 pixels); po = the pixel's index
// in the whole target, for the optional float planes of the tests.  The uniform options are real branches (an empty volatile asm
// keeps the compiler from turning them into conversions + selects executed by every pixel).
__device__ __forceinline__ void store_pixel(const StoreArgs &sa, const uint8_t *out, uint32_t o, uint32_t po, f3 color) {
    if (sa.hdr16) { asm volatile(""); color = mk(through_half(color.x), through_half(color.y), through_half(color.z)); }
    uint32_t word;
    f3 g = mk(0.0f, 0.0f, 0.0f);
    if (sa.debug & 4) { asm volatile(""); word = rgba8_word(color); }   // bit 2: timing only, no post_process
    else {
        g = gamma_exponent(tonemap(color, sa.tm, sa.exposure), sa.inv_gamma);
        word = rgba8_word_of_exponents(g);
    }
    *(uint32_t __attribute__((address_space(1))) *)((char __attribute__((address_space(1))) *)out + o * 4u) = word;
    if (sa.out_ldr) {
        asm volatile("");
        const f3 l = (sa.debug & 4) ? color : mk(__builtin_amdgcn_exp2f(g.x), __builtin_amdgcn_exp2f(g.y), __builtin_amdgcn_exp2f(g.z));
        sa.out_ldr[(size_t)po * 3] = l.x; sa.out_ldr[(size_t)po * 3 + 1] = l.y; sa.out_ldr[(size_t)po * 3 + 2] = l.z;
    }
    if (sa.out_hdr) { asm volatile(""); sa.out_hdr[(size_t)po * 3] = color.x; sa.out_hdr[(size_t)po * 3 + 1] = color.y; sa.out_hdr[(size_t)po * 3 + 2] = color.z; }
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
    return mk(fm(w11, d.x, fm(w01, c.x, fm(w10, b.x, w00 * a.x))), fm(w11, d.y, fm(w01, c.y, fm(w10, b.y, w00 * a.y))),
              fm(w11, d.z, fm(w01, c.z, fm(w10, b.z, w00 * a.z))));
}

// ---- the lights of one lit pixel: get_normal (forward.hlsl:104-111), the sun (:221-222) and the point-light loop (:224-231) ----
// nr, ng, nb: the filtered normal-map channels on the 0..255 scale; gc, gd, ge: world position + tangent frame packed like the
// G-buffer planes c, d, e.  Returns Lo without the (1 - shadow) factor.  Shared by the fast and the general tile below, so a pixel
// gets the same bits whichever of them shades it.
// LOOP 1: scalar loop.  LOOP 2: two lights at a time in packed fp32.  Both read the lights through the scalar cache.
// STATS: count lit pixels, evaluated lights, contributing (n.wi > 0) evaluations and wave-wide zero evaluations into sp.stats.
template <int LOOP, bool STATS>
__device__ __forceinline__ f3 lit_radiance(SP sp, uint32_t lane, float nr, float ng, float nb, float rough, float metal, f3 base,
                                           const float4 &gc, const float4 &gd, const float4 &ge) {
    // get_normal :104-111: rgb with g -> 1 - g, * 2 - 1, then mul(tbn, v), tbn columns t, b, n
    const float r = snorm_of_bytes(nr), g = -snorm_of_bytes(ng), b = snorm_of_bytes(nb);   // (1 - g) * 2 - 1 = -(2 g - 1); nr, ng, nb on the 0..255 scale
    const f3 n = normalize(mk(fm(ge.y, b, fm(gd.z, g, gc.w * r)), fm(ge.z, b, fm(gd.w, g, gd.x * r)), fm(ge.w, b, fm(ge.x, g, gd.y * r))));
    const f3 world = mk(gc.x, gc.y, gc.z);
    const LightArgs la = light_args(sp);   // one batch of scalar loads, in the shadow of the tile's second wave of vector loads
    const f3 wo = normalize(mk(la.eye[0], la.eye[1], la.eye[2]) - world);
    // metalness is only needed after the light loop; left alone the compiler filters it there, and keeps the footprint's texels and
    // weights (12 registers) alive through the loop for it.  The empty asm pins the filtered value in front of the loop.
    asm volatile("" : "+v"(metal));
    LoopPix px;
    TailPix tp;
    make_pix(n, wo, world, base, metal, rough, px, tp);
    unsigned long long contributing = 0ull, wave_zero = 0ull;   // STATS
    f3 Lo;
    if (LOOP == 1) {
        Sums S;
        {   // the sun: wi = -sun_dir, radiance = sun_color (forward.hlsl:221-222)
            const SunArgs sun = sun_args(sp);
            const f3 d = mk(-sun.dir[0], -sun.dir[1], -sun.dir[2]);
            float s1, s2, s3;
            light_scalars<false>(px, d, dot(n, d), s1, s2, s3);
            for (int k = 0; k < 3; ++k) { S.a[k] = sun.color[k] * s1; S.b[k] = sun.color[k] * s2; S.c[k] = sun.color[k] * s3; }
        }
        for (uint32_t i = 0; i < la.n_lights; ++i) {
            // wave-uniform, read through the CONSTANT address space: scalar loads whatever the compiler thinks may have written
            // to global memory before (the volatile asm statements above count as writes, and a plain load would then be a vector load)
            typedef const f4v __attribute__((address_space(4))) *const_f4;
            const f4v lp = ((const_f4)la.lights)[2 * i], lc = ((const_f4)la.lights)[2 * i + 1];
            const f3 dl = mk(lp.x, lp.y, lp.z) - world;
            const float ndl = dot(n, dl);
            if (STATS) { const unsigned long long mk_ = __ballot(ndl > 0.0f); contributing += __popcll(mk_); wave_zero += mk_ == 0ull ? 1 : 0; }
            if (sp.culling && ndl <= 0.0f) continue;      // n.wi <= 0: the term is multiplied by max(n.wi, 0) = 0 (:191-192)
            float s1, s2, s3;
            light_scalars<true>(px, dl, ndl, s1, s2, s3);
            S.a[0] = __builtin_fmaf(lc.x, s1, S.a[0]); S.a[1] = __builtin_fmaf(lc.y, s1, S.a[1]); S.a[2] = __builtin_fmaf(lc.z, s1, S.a[2]);
            S.b[0] = __builtin_fmaf(lc.x, s2, S.b[0]); S.b[1] = __builtin_fmaf(lc.y, s2, S.b[1]); S.b[2] = __builtin_fmaf(lc.z, s2, S.b[2]);
            S.c[0] = __builtin_fmaf(lc.x, s3, S.c[0]); S.c[1] = __builtin_fmaf(lc.y, s3, S.c[1]); S.c[2] = __builtin_fmaf(lc.z, s3, S.c[2]);
        }
        Lo = resolve_sums(tp, S.a, S.b, S.c);
    } else {
        Sums2 S;
        for (int k = 0; k < 3; ++k) { S.a[k] = (v2){0.0f, 0.0f}; S.b[k] = (v2){0.0f, 0.0f}; S.c[k] = (v2){0.0f, 0.0f}; }
        const uint32_t n_pairs = (la.n_lights + 1) >> 1;
        const PackedPix pk = pack_pix(px);
        // The light pairs (3 x 16 bytes per pair: {x0,x1,y0,y1} {z0,z1,r0,r1} {g0,g1,b0,b1}) come through the scalar cache into two
        // sets of SGPRs used alternately: the loads of pair p + 1 are issued before pair p is evaluated (one s_waitcnt per pair,
        // hundreds of cycles after its loads).  Written as asm because the compiler's own version addresses every dword separately
        // (40 scalar instructions per trip) and waits right after issuing.
        const char *lp = reinterpret_cast<const char *>(la.pairs);
        auto finish = [&](v2 dx, v2 dy, v2 dz, const f4v &Bq, const f4v &C, uint32_t p) {
            const v2 nd = accumulate_pair(pk, dx, dy, dz, (v2){Bq.z, Bq.w}, (v2){C.x, C.y}, (v2){C.z, C.w}, S);
            if (STATS) {
                const bool second = 2 * p + 1 < la.n_lights;
                const unsigned long long m0 = __ballot(nd.x > 0.0f), m1 = second ? __ballot(nd.y > 0.0f) : ~0ull;
                contributing += __popcll(m0) + (second ? __popcll(m1) : 0);
                wave_zero += (m0 == 0ull ? 1 : 0) + (m1 == 0ull ? 1 : 0);
            }
        };
        if (n_pairs) {
            f4v A0, B0, C0, A1, B1, C1;
            v2 dx, dy, dz;
            load_light_pair(lp, A0, B0, C0);
            for (uint32_t p = 0;;) {
                wait_and_sub((v2){A0.x, A0.y}, (v2){A0.z, A0.w}, (v2){B0.x, B0.y}, pk.w_xy, pk.wz_a2, dx, dy, dz);   // set 0 has landed
                if (p + 1 < n_pairs) load_light_pair(lp + 48 * (p + 1), A1, B1, C1);                                  // set 1 in flight
                finish(dx, dy, dz, B0, C0, p);
                if (++p == n_pairs) break;
                wait_and_sub((v2){A1.x, A1.y}, (v2){A1.z, A1.w}, (v2){B1.x, B1.y}, pk.w_xy, pk.wz_a2, dx, dy, dz);
                if (p + 1 < n_pairs) load_light_pair(lp + 48 * (p + 1), A0, B0, C0);
                finish(dx, dy, dz, B1, C1, p);
                if (++p == n_pairs) break;
            }
            // No load is in flight here (the last trip issues none), but only the trip counts say so: a path-insensitive reader of the ISA --
            // tools/isa_lint.py follows every asm scalar load along every branch to its wait -- sees the loop's exits behind a load of the
            // other set.  One scalar instruction per lit tile makes "nothing of a light pair is pending behind the loop" a fact of the code.
            asm volatile("s_waitcnt lgkmcnt(0)");
        }
        float A[3], Bs[3], Cs[3];
        {   // the sun joins the sums
            const SunArgs sun = sun_args(sp);
            const f3 d = mk(-sun.dir[0], -sun.dir[1], -sun.dir[2]);
            float s1, s2, s3;
            light_scalars<false>(px, d, dot(n, d), s1, s2, s3);
            for (int k = 0; k < 3; ++k) {
                A[k] = __builtin_fmaf(sun.color[k], s1, S.a[k].x + S.a[k].y);
                Bs[k] = __builtin_fmaf(sun.color[k], s2, S.b[k].x + S.b[k].y);
                Cs[k] = __builtin_fmaf(sun.color[k], s3, S.c[k].x + S.c[k].y);
            }
        }
        Lo = resolve_sums(tp, A, Bs, Cs);
    }
    if (STATS) {
        const unsigned long long active = __ballot(1);
        if (lane == (uint32_t)__ffsll((long long)active) - 1) {
            atomicAdd(sp.stats, (unsigned long long)__popcll(active) * la.n_lights);   // point-light evaluations of lit pixels
            atomicAdd(sp.stats + 1, (unsigned long long)__popcll(active));            // lit pixels
            atomicAdd(sp.stats + 2, contributing);                                     // ... of which n.wi > 0
            atomicAdd(sp.stats + 3, wave_zero);                                        // (tile, light) pairs with n.wi <= 0 in every lit lane
            atomicAdd(sp.stats + 4, 1ull);                                             // tiles with a lit pixel
        }
    }
    return Lo;
}

// ---- ps_main + post_process for one 8x8 tile (one wave): the FAST tile ---------------------------------------------------
// What nearly every tile of a frame is: wholly inside the target, every pixel covered by ONE material stored packed, and the
// shadow test of every pixel decided by the bounds table (or no shadow map at all).  For such a tile the wave runs straight-line
// code with wave-uniform branches only -- material descriptor in SGPRs, the footprint as two 16-byte loads, the RGBA8 target
// addressed from a scalar base -- except for the lit pixels' part.  Returns false, having stored nothing, when the tile is not of
// that kind (decided wave-wide; the texel loads already issued are then dropped): the caller shades it with shade_tile.
// `second(pc, pd, pe, gc, gd, ge)` delivers the lit pixels' remaining attributes (world position + tangent frame, packed like the
// G-buffer planes c, d, e): loaded from the G-buffer, or interpolated on the spot by the visibility-buffer kernel.
template <int LOOP, bool STATS, bool PCF_CAND, class Second>
__device__ __forceinline__ bool shade_tile_fast(SP sp, KernArgs args, const ArgsA &A, const float *lut, uint32_t ty, uint32_t tx, uint32_t lane, const TileHead &cur, Second second) {
    ArgsB B = args_b(args);   // (the head of the tile is in flight)
    const int32_t row0 = (int32_t)(ty * 8) - (int32_t)B.row0_in_tile;   // the tile's first pixel row in the target (wave-uniform)
    if ((A.debug & (1 | 2 | 4 | 256)) != 0 || tx * 8 + 8 > B.width || row0 < 0 || row0 + 8 > (int32_t)B.rows) return false;
    if (B.sh.map != nullptr && B.sh.bounds == nullptr) return false;
    const uint32_t mat = __float_as_uint(cur.b2);
    const uint32_t m0 = __builtin_amdgcn_readfirstlane(mat);
    if (m0 >= A.n_materials || __ballot(mat != m0) != 0ull) return false;
    u8v dv;
    float lsx = cur.a.z;
    tex_desc_issue(B.tex, m0 * 3, dv, lsx);   // the material's descriptor: in flight over the shadow coordinates
    // ---- B: shadow test, forward.hlsl:68-96, from the bounds table alone: the entry's load goes out first (it decides whether the
    // tile's second wave of loads is needed, and loads come back in the order they were asked for)
    ShadowPos spos;
    float2 mm;
    uint32_t toff = 0;
    const bool q8s = (A.debug & SAMPLER_Q8_SHADOW) != 0;   // (wave-uniform: ARCTIC_OPT_SAMPLER)
    const bool in_table = shadow_quick_issue(B.sh, lsx, cur.a.w, cur.b0, cur.b1, spos, mm, &toff, q8s);
    // ---- A: material fetch, forward.hlsl:98-124: two 16-byte texel loads
    TexS d0 = tex_desc_wait(dv, toff);
    if (!d0.packed) return false;
    Taps pt;
    fetch_taps_packed(d0, cur.a.x, cur.a.y, pt, (A.debug & SAMPLER_Q8_MATERIAL) != 0);
    ArgsC C = args_c(args, A, B);   // the next batch of arguments arrives in the shadow of those loads
    float lit;
    const bool decided = shadow_quick_decide(B.sh, in_table, spos, mm, lit);
    if (__ballot(!decided) != 0ull) {   // a tile on a shadow edge (a tenth of the benchmark frame's tiles), or at the map's border
        // Round 4: the 25 taps of the undecided pixels are taken here, with the texel loads above still in flight, instead of handing the
        // tile to the general code, which started over (descriptor, texels, table entry -- three more memory round trips).
#if !ARCTIC_EDGE_IN_FAST
        return false;
#endif
        if (B.sh.S > 5000u) return false;   // (no 4x4 window for such maps: the general tile)
        // the 25-tap code is what would set the kernel's register count: the footprint's texels are given up across it (an empty asm
        // "writes" every component) and asked for again behind it -- cache hits, and such tiles are few
        asm("" : "=v"(pt.r0.x), "=v"(pt.r0.y), "=v"(pt.r0.z), "=v"(pt.r0.w), "=v"(pt.r1.x), "=v"(pt.r1.y), "=v"(pt.r1.z), "=v"(pt.r1.w));
        asm("" : "=v"(pt.w00), "=v"(pt.w10), "=v"(pt.w01), "=v"(pt.w11));
        bool tapped = false;
        if (q8s) { asm volatile(""); if (!decided) lit = 1.0f - shadow_window<false, true>(B.sh.map, B.sh.S, spos.px, spos.py, spos.pz); }   // (no statistics of this one)
        else if (!decided) lit = 1.0f - shadow_window<PCF_CAND>(B.sh.map, B.sh.S, spos.px, spos.py, spos.pz, STATS ? &tapped : nullptr);
        if (STATS) {   // [5] tiles with a pixel the table left undecided, [6] such pixels, [7] tiles that ran the 25 compares, [8] pixels that did
            const unsigned long long und = __ballot(!decided), tap = __ballot(tapped);
            if (wave_lane() == 0) {
                atomicAdd(sp.stats + 5, 1ull); atomicAdd(sp.stats + 6, (unsigned long long)__popcll(und));
                if (tap) { atomicAdd(sp.stats + 7, 1ull); atomicAdd(sp.stats + 8, (unsigned long long)__popcll(tap)); }
            }
        }
        // ... and so are the argument batches and the descriptor (the 25 taps hold a dozen lane masks in scalar registers: with the
        // batches alive across them the kernel would pass 96 SGPRs, i.e. lose a wave per SIMD): loaded again, two scalar round trips
        asm volatile("" : "+s"(args));
        B = args_b(args);
        C = args_c(args, A, B);
        d0 = tex_desc(B.tex, m0 * 3);
        fetch_taps_packed(d0, cur.a.x, cur.a.y, pt, (A.debug & SAMPLER_Q8_MATERIAL) != 0);
    }
    // exact culling: Lo of ps_main is a sum of terms each multiplied by (1 - shadow) (point lights too: forward.hlsl:222,
    // 230), so a fully shadowed pixel is ambient * base and needs neither the sun, nor any point light, nor its normal,
    // tangent frame, position, metalness or roughness.
    const bool live = B.culling ? lit != 0.0f : true;
    float4 gc, gd, ge;
    if (live) second(C.gc, C.gd, C.ge, gc, gd, ge);   // second wave of loads: lit pixels only (48 B / pixel, whole 128-byte tile rows)
    // ---- C: base colour
    f3 base = mk(filt_srgb<0>(pt, lut), filt_srgb<1>(pt, lut), filt_srgb<2>(pt, lut));
#if ARCTIC_PIN_SECOND_WAVE
    // Round 5, read off the ISA: the compiler copied components of gc / gd / ge into the register pairs its packed multiplies want RIGHT BEHIND the
    // three loads (s_waitcnt vmcnt(2), (1), (0) + v_mov inside the `live` branch): the wave sat out the whole memory latency there, in front of the
    // twelve table look-ups of the base colour.  The twelve values and a component of the base colour pass through one empty asm statement: whatever
    // the compiler does with the loaded registers, it does behind the base colour.
    asm volatile("" : "+v"(gc.x), "+v"(gc.y), "+v"(gc.z), "+v"(gc.w), "+v"(gd.x), "+v"(gd.y), "+v"(gd.z), "+v"(gd.w), "+v"(ge.x), "+v"(ge.y), "+v"(ge.z), "+v"(ge.w), "+v"(base.x));
#endif
    // ---- D: the lights.  (The ambient term is formed behind the light loop on either side of the branch: formed in front of it, it
    // would occupy three registers through the loop.)
    EpiArgs E = {C.st, B.out, B.width, B.row0_in_tile, C.ambient};
    f3 color;
    if (__ballot(live) != 0ull) {   // (wave-uniform: the epilogue's arguments stay scalar on either side)
        if (live) {
            const f3 Lo = lit_radiance<LOOP, STATS>(sp, lane, filt_bytes<0, 3>(pt), filt_bytes<1, 0>(pt), filt_bytes<1, 1>(pt),
                                                    filt_unorm<1, 2>(pt), filt_unorm<1, 3>(pt),   // metal-rough .g, .b (forward.hlsl:117,123)
                                                    base, gc, gd, ge);
            color = Lo;
        }
        asm volatile("" : "+s"(args));
        E = epi_args(args);
        const f3 amb = base * E.ambient;
        if (live) color = mk(__builtin_fmaf(color.x, lit, amb.x), __builtin_fmaf(color.y, lit, amb.y), __builtin_fmaf(color.z, lit, amb.z));
        else color = amb;
    } else color = base * E.ambient;
    // ---- E: post_process + store: the tile's first pixel is a scalar address, the lane adds (lane >> 3) rows + (lane & 7)
    // (readfirstlane: on a value in an SGPR a scalar move.  With the tiled-texture branch in fetch_taps_packed the compiler carried these two through the
    //  tile in VECTOR registers -- and did the tile's address arithmetic below with vector instructions, 13 per tile: round 5, found with SQ_INSTS_VALU)
    const uint32_t e_width = __builtin_amdgcn_readfirstlane(E.width), e_row0 = __builtin_amdgcn_readfirstlane(E.row0_in_tile);
    const uint32_t tile_px = (ty * 8 - e_row0) * e_width + tx * 8;   // (the tile is wholly inside the target: ty * 8 >= row0_in_tile)
    const uint32_t l2 = wave_lane();
    const uint32_t o = __umul24(l2 >> 3, e_width) + (l2 & 7u);   // (width <= 16384)
    store_pixel(E.st, E.out + (size_t)tile_px * 4u, o, tile_px + o, color);
    return true;
}

// ---- the GENERAL tile: ragged tiles at the target's edge, pixels without geometry (skybox), several materials in one tile,
// materials with images of unequal sizes, the 25-tap shadow test, the debug / timing options --------------------------------
template <int LOOP, bool STATS, bool LDS_SHADOW, class Second>
__device__ __forceinline__ void shade_tile(SP sp, const float *lut, float *shadow_tile, uint32_t ty, uint32_t tx,
                                           uint32_t lane, const TileHead &cur, Second second) {
    const uint32_t x = tx * 8 + (lane & 7);
    const int32_t y = (int32_t)(ty * 8 + (lane >> 3)) - (int32_t)sp.row0_in_tile;
    const bool in_frame = x < sp.width && y >= 0 && y < (int32_t)sp.rows;
    const uint32_t mat = __float_as_uint(cur.b2);
    const bool covered = in_frame && mat < sp.n_materials;
    const uint32_t o = (uint32_t)y * sp.width + x;   // targets are at most 16384^2 pixels
    const float u = cur.a.x, v = cur.a.y;
    const bool q8m = (sp.debug & SAMPLER_Q8_MATERIAL) != 0, q8s = (sp.debug & SAMPLER_Q8_SHADOW) != 0;   // ARCTIC_OPT_SAMPLER (wave-uniform)

    // ---- A: material fetch, forward.hlsl:98-124.  Texel loads of packed materials stay in flight over the shadow test.
    Taps pt;
    f3 base = mk(0.0f, 0.0f, 0.0f);
    bool plain = false;   // lane's material is stored as three plain RGBA8 images (unequal sizes): the cold path
    auto fetch_material = [&]() {
        // an empty asm "writes" every component: nothing of an earlier fetch stays live across the shadow test's slow path
        asm("" : "=v"(pt.r0.x), "=v"(pt.r0.y), "=v"(pt.r0.z), "=v"(pt.r0.w), "=v"(pt.r1.x), "=v"(pt.r1.y), "=v"(pt.r1.z), "=v"(pt.r1.w));
        asm("" : "=v"(pt.w00), "=v"(pt.w10), "=v"(pt.w01), "=v"(pt.w11));
        if (sp.debug & 1) {   // timing only: no texture traffic
            pt.r0 = pt.r1 = (u4v){0x808080u, 0u, 0x808080u, 0u}; pt.w00 = pt.w10 = pt.w01 = pt.w11 = 0.25f;
            return;
        }
        auto fetch = [&](uint32_t m, bool mine) {   // m wave-uniform: the descriptor comes through the scalar unit
            const TexS d0 = tex_desc(sp.tex, m * 3);
            if (mine) {
                if (d0.packed) fetch_taps_packed(d0, u, v, pt, q8m);
                else {
                    Taps t0;
                    fetch_taps_plain(d0, u, v, t0, q8m);
                    base = mk(filt_srgb<0>(t0, lut), filt_srgb<1>(t0, lut), filt_srgb<2>(t0, lut));
                    plain = true;
                }
            }
        };
        unsigned long long todo = __ballot(covered);
        if (todo == 0ull) return;
        const uint32_t m0 = __builtin_amdgcn_readlane(mat, __ffsll((long long)todo) - 1);
        if (__ballot(covered && mat != m0) == 0ull) {   // one material in the tile: straight-line code, the texel
            fetch(m0, covered);                        // loads go straight into their registers
            return;
        }
        while (todo) {   // a waterfall loop over the distinct materials of a mixed tile
            const uint32_t m = __builtin_amdgcn_readlane(mat, __ffsll((long long)todo) - 1);
            const bool mine = covered && mat == m;
            fetch(m, mine);
            todo &= ~__ballot(mine);
        }
    };
    fetch_material();

    // ---- B: shadow test, forward.hlsl:68-96 ----------------------------------------------------------------------
    float lit = 1.0f;
    if (!(sp.debug & 2)) {
        ShadowPos spos;
        const ShadowArgs sh = shadow_args(sp);
        const bool decided = !covered || shadow_quick(sh, cur.a.z, cur.a.w, cur.b0, cur.b1, spos, lit, q8s);
        if (__ballot(!decided) != 0ull) {   // a tile on a shadow edge (or at the map's border)
            const bool staged = LDS_SHADOW && !q8s && sh.S <= 5000u && shadow_lds_tile(sh, shadow_tile, lane, !decided, spos.px, spos.py, spos.pz, lit);
            if (!staged && !decided) lit = shadow_slow(sh, spos, q8s);
            // the 25-tap path is what sets the kernel's register count: the texels fetched above are dropped across it and
            // fetched again (cache hits; such tiles are few) instead of being kept alive through it
            fetch_material();
        }
    }
    // exact culling: see shade_tile_fast
    const bool live = covered && (sp.culling ? lit != 0.0f : true);
    float4 gc, gd, ge;
    if (live) second(sp.g.c, sp.g.d, sp.g.e, gc, gd, ge);   // second wave of loads: lit pixels only (48 B / pixel, whole 128-byte tile rows)

    // ---- C: base colour; pixels without geometry: the skybox -----------------------------------------------------------
    if (covered && !plain) base = mk(filt_srgb<0>(pt, lut), filt_srgb<1>(pt, lut), filt_srgb<2>(pt, lut));
    f3 color = base * sp.ambient;
    if (in_frame && !covered && sp.env) {
        const int gy = (row_global((int)ty, sp.band_tiles, sp.shard_count, sp.shard_index) + sp.tile_y0) * 8 + (int)(lane >> 3);
        const float nx = __builtin_fmaf((float)x + 0.5f, sp.ndc_sx, -1.0f), ny = __builtin_fmaf(-((float)gy + 0.5f), sp.ndc_sy, 1.0f);
        color = sample_environment(sp.env, sp.env_w, sp.env_h,
                                   __builtin_fmaf(sp.sky_up[0], ny, __builtin_fmaf(sp.sky_right[0], nx, sp.sky_fwd[0])),
                                   __builtin_fmaf(sp.sky_up[1], ny, __builtin_fmaf(sp.sky_right[1], nx, sp.sky_fwd[1])),
                                   __builtin_fmaf(sp.sky_up[2], ny, __builtin_fmaf(sp.sky_right[2], nx, sp.sky_fwd[2])));
    }

    // ---- D: the lights ---------------------------------------------------------------------------------------------
    if (live) {
        float nr, ng, nb, rough, metal;
        if (!plain) {
            nr = filt_bytes<0, 3>(pt); ng = filt_bytes<1, 0>(pt); nb = filt_bytes<1, 1>(pt);
            rough = filt_unorm<1, 2>(pt); metal = filt_unorm<1, 3>(pt);   // metal-rough .g, .b (forward.hlsl:117,123)
        }
        if (__ballot(plain) != 0ull) {   // cold: normal and metal-rough images of the plain materials, one material at a time
            unsigned long long todo = __ballot(plain);
            while (todo) {
                const uint32_t m = __builtin_amdgcn_readlane(mat, __ffsll((long long)todo) - 1);
                const bool mine = plain && mat == m;
                const TexS d1 = tex_desc(sp.tex, m * 3 + 1), d2 = tex_desc(sp.tex, m * 3 + 2);
                if (mine) {
                    Taps t1, t2;
                    fetch_taps_plain(d1, u, v, t1, q8m);
                    fetch_taps_plain(d2, u, v, t2, q8m);
                    nr = filt_bytes<0, 0>(t1); ng = filt_bytes<0, 1>(t1); nb = filt_bytes<0, 2>(t1);
                    rough = filt_unorm<0, 1>(t2); metal = filt_unorm<0, 2>(t2);
                }
                todo &= ~__ballot(mine);
            }
        }
        const f3 Lo = lit_radiance<LOOP, STATS>(sp, lane, nr, ng, nb, rough, metal, base, gc, gd, ge);
        color = mk(__builtin_fmaf(Lo.x, lit, color.x), __builtin_fmaf(Lo.y, lit, color.y), __builtin_fmaf(Lo.z, lit, color.z));
    }

    // ---- E: post_process + store ---------------------------------------------------------------------------------------
    if (in_frame) store_pixel(store_args(sp), sp.out_rgba8, o, o, color);
}

// ARCTIC_OPT_TILE_TRACE (a measuring aid, off by default: one wave-uniform branch at either end of a tile): when and where every tile
// was shaded.  Per tile 4 x u64: s_memrealtime (the 100 MHz reference clock, the same on every XCD) at the start and the end of its
// wave's work, HW_ID | XCC_ID << 32 (which XCD / SE / CU / SIMD / wave slot) | reference-clock ticks between the kernel's entry and the start << 40, and 1 = the fast tile | shader-clock ticks (s_memtime)
// between start and end << 8.  tools/experiments/tile_trace.py turns it into per-SIMD timelines.
// Nothing of it lives in registers across a tile (six scalar registers through the whole kernel, when it did): the start stamps go to the
// trace buffer at once and trace_end reads them back.  `entry`: the wave's first tile only (0: a later tile of the wave).
__device__ __forceinline__ unsigned long long trace_entry() { return __builtin_amdgcn_s_memrealtime(); }   // first thing in the kernel (unconditional: asking whether anyone traces would be a scalar round trip in front of everything)
__device__ __forceinline__ void trace_begin(SP sp, size_t tile, unsigned long long entry) {   // in front of a tile's work (the pointer's load: in the shadow of the tile's first bytes.
    if (!sp.trace) return;                                                                       //  Asking ArgsA::debug instead trips the compiler: "illegal VGPR to SGPR copy" in the scalar-loop kernels)
    const unsigned long long real = __builtin_amdgcn_s_memrealtime(), core = __builtin_amdgcn_s_memtime();
    if (wave_lane() == 0) { unsigned long long *o = sp.trace + tile * 4; o[0] = real; o[2] = (entry ? real - entry : 0ull) << 40; o[3] = core; }
}
__device__ __forceinline__ void trace_end(SP sp, const ArgsA &A, size_t tile, bool fast) {
    if (!(A.debug & DEBUG_TRACE)) return;
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long hw = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32);   // HW_REG_HW_ID, HW_REG_XCC_ID
    if (wave_lane() == 0) { unsigned long long *o = sp.trace + tile * 4; o[1] = r1; o[2] |= hw; o[3] = (fast ? 1ull : 0ull) | ((c1 - o[3]) << 8); }
}

// the kernel's argument block (its only parameter, at offset 0 of the kernarg segment)
__device__ __forceinline__ KernArgs kernel_args() { return (KernArgs)__builtin_amdgcn_kernarg_segment_ptr(); }

// LDS: the sRGB LUT, 1 KiB per workgroup.  EVERY wave writes all of it (16 bytes per lane: the four waves of a workgroup store the same
// values to the same addresses) and reads it back only behind its own stores -- LDS operations of one wave complete in order -- so no
// barrier stands between a wave's launch and its first tile, and the table's load travels beside the tile's first bytes instead of in
// front of them (round 3: argument loads -> table load -> barrier -> head loads, 1.1-1.4 us of every wave's life).
struct LutRegs { float4 v; };
__device__ __forceinline__ LutRegs lut_load(const float *srgb_lut, uint32_t lane) { LutRegs r; r.v = gload_f4(srgb_lut, lane * 16u); return r; }
__device__ __forceinline__ void lut_store(float *lut, uint32_t lane, const LutRegs &r) { reinterpret_cast<float4 *>(lut)[lane] = r.v; }

// ---- the shading pass over a resident G-buffer ------------------------------------------------------------------------
// Which tiles a wave shades.  Lit regions (ALU-bound tiles: ~8000 issue cycles each at 64 lights) and shadowed ones (latency-bound:
// ~150 instructions behind three memory round trips) are spatially clustered -- in the benchmark scene the sun reaches the lower
// 40 % of the frame -- and the hardware dispatches workgroups in order and deals them to the shader engines round-robin whatever
// their load.  One tile per wave, top to bottom, makes the launch latency-bound first and ALU-bound afterwards; handing out the
// two ends of the frame alternately makes half the shader engines the one and half the other (profiles/r3_tile_trace_*.txt).
// So every wave shades T tiles, one after the other, 1 / T of the frame's height apart (ARCTIC_OPT_TILES_PER_WAVE): every wave
// then carries the same mix, at any moment the waves of a SIMD are spread over both kinds, and each kind hides behind the other
// wherever in the frame the light falls; and a wave is launched once for T tiles.
// XCD-aware: workgroups are dealt round-robin over the 8 XCDs (each with its own 4 MiB L2), so physical block b runs on XCD b % 8.
// A workgroup = 4 horizontally adjacent tiles (they share texture and shadow-map lines), T times; XCD x takes the tile rows
// y = x (mod 8), walking each row left to right.  grid = (8 x workgroups per tile row, 1 / T of the groups of 8 tile rows): the
// linear block id advances along x first, so id % 8 = x % 8.  (Placement is a speed matter only; surplus blocks exit.)
// WITH A DISPATCH ORDER (round 4; ShadeParams::tile_order, written by the G-buffer prepass: geometry.hip k_tile_order) the grid is one-
// dimensional and block b takes the jobs b T ... b T + T - 1 of the list, a job = a strip of 4 horizontally adjacent tiles: the
// prepass knows which tiles can be lit at all, and deals those evenly over the dispatch, the last stretch excepted.
// The wave's next tile from job k on: true with (tx, ty) and k = the job taken; false: no more work for this wave.
// Workgroups of ONE wave (ARCTIC_WG_WAVES == 1): the four tiles of a strip are the blocks b, b + 8, b + 16, b + 24 of a group of 32 -- the
// same XCD (blocks are dealt round-robin over the 8 XCDs), so what the strip's tiles share still meets in one L2 -- and the strip is
// block (b >> 5) * 8 + (b & 7) of the 4-wave numbering below.
struct BlockId { uint32_t x, y, wave; };
template <int WGW>
__device__ __forceinline__ BlockId block_id() {
    BlockId b;
    if (WGW == 1) { b.x = (blockIdx.x >> 5) * 8 + (blockIdx.x & 7u); b.wave = (blockIdx.x >> 3) & 3u; }
    else { b.x = blockIdx.x; b.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); }
    b.y = blockIdx.y;
    return b;
}
__device__ __forceinline__ bool next_tile(const ArgsA &A, const OrderArgs &O, const BlockId &b, uint32_t &k, uint32_t &tx, uint32_t &ty) {
    if (O.order) {
        for (; k < A.T; ++k) {
            const uint32_t j = b.x * A.T + k;
            if (j >= O.n_jobs) return false;
            const uint32_t e = O.order[j];
            ty = e >> 16; tx = (e & 0xFFFFu) * 4 + b.wave;
            if (tx < A.tiles_x && ty < A.tiles_y) return true;   // (a strip at the right edge may hold fewer than 4 tiles)
        }
        return false;
    }
    tx = (b.x >> 3) * 4 + b.wave;
    if (k >= A.T || tx >= A.tiles_x) return false;
    const uint32_t g = b.y + k * A.stride;   // the block's groups of 8 tile rows: blockIdx.y + k stride
    ty = g * 8 + (b.x & 7u);
    return g < ((A.tiles_y + 7u) >> 3) && ty < A.tiles_y;
}
template <int LOOP, bool STATS, bool LDS_SHADOW>
__global__ __launch_bounds__(64 * ARCTIC_WG_WAVES) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_material(const ShadeParams sp_by_value) {
    __shared__ float lut[256];
    __shared__ float shadow_tiles[LDS_SHADOW ? ARCTIC_WG_WAVES : 1][LDS_SHADOW ? SHADOW_TILE * SHADOW_TILE : 1];   // one per wave (the LDS variant of the PCF slow path)
    KernArgs args = kernel_args();
    unsigned long long t_entry = trace_entry();
    const BlockId blk = block_id<ARCTIC_WG_WAVES>();
    const uint32_t wave = ARCTIC_WG_WAVES == 1 ? 0u : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (of the workgroup: which LDS shadow tile is this wave's)
    const float *srgb_lut;
    const unsigned long long *vis_unused;
    OrderArgs O;
    ArgsA A = args_a_first(args, srgb_lut, vis_unused, O);
    uint32_t tx, ty, k = 0;
    TileHead cur;
#if ARCTIC_LUT_SHARED && ARCTIC_WG_WAVES == 4
    {   // a quarter of the table per wave (waves without a tile load theirs too), the barrier behind the head loads' issue
        const bool has_tile = next_tile(A, O, blk, k, tx, ty);
        const uint32_t lane = wave_lane();
        const float q = __uint_as_float(gload_u32(srgb_lut, (blk.wave * 64u + lane) * 4u));
        if (has_tile) cur = load_head(A.ga, A.gb, (size_t)ty * A.tiles_x + tx, lane);
        lut[blk.wave * 64u + lane] = q;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (!has_tile) return;
    }
#else
    if (!next_tile(A, O, blk, k, tx, ty)) return;
    {
        const uint32_t lane = wave_lane();
        const LutRegs lr = lut_load(srgb_lut, lane);                                       // an L2 hit: back first ...
        cur = load_head(A.ga, A.gb, (size_t)ty * A.tiles_x + tx, lane);                   // ... while the tile's first bytes travel
        lut_store(lut, lane, lr);
    }
#endif
#pragma nounroll
    for (;;) {
        asm volatile("" : "+s"(args));   // see SP
        SP sp = *args;
        const uint32_t lane = wave_lane();
        const size_t tile = (size_t)ty * A.tiles_x + tx;   // wave-uniform
        trace_begin(sp, tile, t_entry);
        t_entry = 0ull;
        const auto second = [&](const float4 *pc, const float4 *pd, const float4 *pe, float4 &gc, float4 &gd, float4 &ge) {
            gc = gload_f4(pc + tile * 64, lane * 16u); gd = gload_f4(pd + tile * 64, lane * 16u); ge = gload_f4(pe + tile * 64, lane * 16u);
        };
        const bool fast = shade_tile_fast<LOOP, STATS, ARCTIC_PCF_CANDIDATES != 0>(sp, args, A, lut, ty, tx, lane, cur, second);
        if (!fast) shade_tile<LOOP, STATS, LDS_SHADOW>(sp, lut, shadow_tiles[LDS_SHADOW ? wave : 0], ty, tx, lane, cur, second);
        trace_end(sp, A, tile, fast);
        if (++k >= A.T) break;
        asm volatile("" : "+s"(args));
        A = args_a(args);                      // (nothing of the block stays in registers across a tile)
        O = order_args(args);
        if (!next_tile(A, O, blk, k, tx, ty)) break;
        cur = load_head(A.ga, A.gb, (size_t)ty * A.tiles_x + tx, wave_lane());
    }
}

// ---- the same without a G-buffer (whole frames): the tile walk straight from the visibility plane ----------------------
// arctic_render_frame has no use for the 76 B/pixel G-buffer between its own two kernels: writing it (k_resolve, 630 MB at
// 4K) and reading it back costs more than interpolating again.  This variant reads the 8-byte visibility key, finds the
// triangle, and interpolates uv + light-space position for every covered pixel and world position + tangent frame only
// for the lit ones -- with the very operations of k_resolve (edges.h, fp contraction off), so the pixels are bit-identical
// to the G-buffer path.  Everything after the attributes is shade_tile, shared.
template <int LOOP, bool STATS, bool LDS_SHADOW>
__global__ __launch_bounds__(64 * VIS_WG_WAVES) __attribute__((amdgpu_waves_per_eu(6, 6))) void k_material_vis(const ShadeParams sp_by_value) {
    __shared__ float lut[256];
    __shared__ float shadow_tiles[LDS_SHADOW ? VIS_WG_WAVES : 1][LDS_SHADOW ? SHADOW_TILE * SHADOW_TILE : 1];   // one per wave (the LDS variant of the PCF slow path)
    KernArgs args = kernel_args();
    unsigned long long t_entry = trace_entry();
    const BlockId blk = block_id<VIS_WG_WAVES>();
    const uint32_t wave = VIS_WG_WAVES == 1 ? 0u : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (of the workgroup: which LDS shadow tile is this wave's)
    const float *srgb_lut;     // XCD-aware order or the prepass's dispatch order, T tiles per wave, the LUT without a barrier: see k_material
    const unsigned long long *vis_plane;
    OrderArgs O;
    ArgsA A = args_a_first(args, srgb_lut, vis_plane, O);
    uint32_t tx, ty, k = 0;
    unsigned long long key = ~0ull;
#if ARCTIC_LUT_SHARED
    {
        const bool has_tile = next_tile(A, O, blk, k, tx, ty);
        const uint32_t lane = wave_lane();
        const float q = __uint_as_float(gload_u32(srgb_lut, (blk.wave * 64u + lane) * 4u));
        if (has_tile) key = vis_plane[((size_t)ty * A.tiles_x + tx) * 64 + lane];
        lut[blk.wave * 64u + lane] = q;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (!has_tile) return;
    }
#else
    if (!next_tile(A, O, blk, k, tx, ty)) return;
    {
        const uint32_t lane = wave_lane();
        const LutRegs lr = lut_load(srgb_lut, lane);
        key = vis_plane[((size_t)ty * A.tiles_x + tx) * 64 + lane];
        lut_store(lut, lane, lr);
    }
#endif
#pragma nounroll
    for (;;) {
    asm volatile("" : "+s"(args));   // see SP
    SP sp = *args;
    const uint32_t lane = wave_lane();
    trace_begin(sp, (size_t)ty * A.tiles_x + tx, t_entry);
    t_entry = 0ull;
    const int32_t px = (int32_t)(tx * 8 + (lane & 7));
    const int32_t py = (row_global((int)ty, sp.band_tiles, sp.shard_count, sp.shard_index) + sp.tile_y0) * 8 + (int32_t)(lane >> 3);
    TileHead cur;
    cur.a = make_float4(0.0f, 0.0f, 0.0f, 0.0f); cur.b0 = 0.0f; cur.b1 = 0.0f; cur.b2 = __uint_as_float(NO_MATERIAL);
    float B[3] = {0.0f, 0.0f, 0.0f};
    uint32_t v0 = 0, v1 = 0, v2 = 0;   // the source triangle's transformed vertices (indices, not pointers: they stay live across the tile)
    // attribute k of transformed vertex v: XVert::attr at byte 16 + 4k of a 96-byte record.  With compact tables (below 4 GiB each, the
    // host says) every gather is a wave-uniform base + a 32-bit byte offset: no 64-bit multiply-adds per lane and load
    const bool compact = sp.compact_tables != 0;
    if (key != ~0ull && compact) {
        const uint32_t ri = gload_u32(sp.rec_of, (uint32_t)key << 2);   // low word of the key = order id (k_setup)
        const uint32_t so = ri << 7;                                     // SetupRec and RasterRec are 128 bytes
        const float4u qf = gload_f4u(sp.rrecs, so + 96u);                // dz2, inv_area, order id, flags
        const u4v src = gload_u4u(sp.rrecs, so + 112u);                  // the source triangle's three transformed vertices, its material
        if (__float_as_uint(qf.w) & RASTER_EXACT_F64) {
            typedef double d2v __attribute__((ext_vector_type(2)));
            const auto ld2 = [&](uint32_t o) { return *(const d2v __attribute__((address_space(1))) *)((gchar)sp.rrecs + o); };
            const auto ld1 = [&](uint32_t o) { return *(const double __attribute__((address_space(1))) *)((gchar)sp.rrecs + o); };
            const double A0 = ld1(so), C0 = ld1(so + 48u), A2 = ld1(so + 16u), B0 = ld1(so + 24u), B2 = ld1(so + 40u), C2 = ld1(so + 64u);
            (void)ld2;
            const float4u s2 = gload_f4u(sp.recs, so + 32u);             // z[2], iw[0..2]
            const uint32_t fl = __float_as_uint(qf.w);
            {   // source_barycentrics (edges.h), on the fields just loaded: the same operations in the same order
#pragma clang fp contract(off)
                const double x = (double)px, y = (double)py;
                const float l1 = (float)__builtin_fma(A2, x, __builtin_fma(B2, y, C2)) * qf.y;
                const float l2 = (float)__builtin_fma(A0, x, __builtin_fma(B0, y, C0)) * qf.y;
                const float l0 = (1.0f - l1) - l2;
                const float pw0 = l0 * s2.y, pw1 = l1 * s2.z, pw2 = l2 * s2.w;
                const float rr = 1.0f / ((pw0 + pw1) + pw2);
                const float c0 = pw0 * rr, c1 = pw1 * rr, c2 = pw2 * rr;
                if (__ballot((fl & RASTER_UNIT_BARY) == 0u) == 0ull) {
                    // every record under the tile is an uncut source triangle (nearly every tile): the rows of its barycentric matrix are
                    // unit vectors and the products below return c0, c1, c2 themselves -- in source order when set-up exchanged two vertices
                    const bool swapped = (fl & RASTER_SWAPPED) != 0u;
                    B[0] = c0; B[1] = swapped ? c2 : c1; B[2] = swapped ? c1 : c2;
                } else {   // a tile with a cut triangle under it: the product per lane, and the lanes of uncut triangles as above (source_barycentrics' rule: edges.h)
                    const float4u b0 = gload_f4u(sp.recs, so + 48u), b1 = gload_f4u(sp.recs, so + 64u);   // bary[0][0..2], bary[1][0] | bary[1][1..2], bary[2][0..1]
                    const float b22 = __uint_as_float(gload_u32(sp.recs, so + 80u));
                    const bool unit = (fl & RASTER_UNIT_BARY) != 0u, swapped = (fl & RASTER_SWAPPED) != 0u;
                    const float p0 = (c0 * b0.x + c1 * b0.w) + c2 * b1.z, p1 = (c0 * b0.y + c1 * b1.x) + c2 * b1.w, p2 = (c0 * b0.z + c1 * b1.y) + c2 * b22;
                    B[0] = unit ? c0 : p0; B[1] = unit ? (swapped ? c2 : c1) : p1; B[2] = unit ? (swapped ? c1 : c2) : p2;
                }
            }
        } else source_barycentrics(sp.recs[ri], sp.rrecs[ri], px, py, B);   // rare: coordinates of 2^24 and more
        cur.b2 = __uint_as_float(src.w);
        v0 = src.x; v1 = src.y; v2 = src.z;
        const uint32_t a0 = v0 * 96u + 16u, a1 = v1 * 96u + 16u, a2 = v2 * 96u + 16u;
        const float2 u0 = gload_f2(sp.xv, a0), u1 = gload_f2(sp.xv, a1), u2 = gload_f2(sp.xv, a2);                    // attr 0, 1
        const float4u w0 = gload_f4u(sp.xv, a0 + 56u), w1 = gload_f4u(sp.xv, a1 + 56u), w2 = gload_f4u(sp.xv, a2 + 56u);   // attr 14..17
        const auto mix = [&](float x0, float x1, float x2) {
#pragma clang fp contract(off)
            return (B[0] * x0 + B[1] * x1) + B[2] * x2;   // interpolate_attr (edges.h)
        };
        cur.a = make_float4(mix(u0.x, u1.x, u2.x), mix(u0.y, u1.y, u2.y), mix(w0.x, w1.x, w2.x), mix(w0.y, w1.y, w2.y));
        cur.b0 = mix(w0.z, w1.z, w2.z); cur.b1 = mix(w0.w, w1.w, w2.w);
    } else if (key != ~0ull) {
        const uint32_t ri = sp.rec_of[(uint32_t)key];
        const SetupRec &t = sp.recs[ri];
        source_barycentrics(t, sp.rrecs[ri], px, py, B);
        const ObjectRec &ob = sp.objs[t.object];
        const uint32_t lt = t.src_tri - ob.first_triangle;
        v0 = ob.first_xvert + ob.indices[3 * lt]; v1 = ob.first_xvert + ob.indices[3 * lt + 1]; v2 = ob.first_xvert + ob.indices[3 * lt + 2];
        const float *A0 = sp.xv[v0].attr, *A1 = sp.xv[v1].attr, *A2 = sp.xv[v2].attr;
        cur.a = make_float4(interpolate_attr(B, A0, A1, A2, 0), interpolate_attr(B, A0, A1, A2, 1),
                            interpolate_attr(B, A0, A1, A2, 14), interpolate_attr(B, A0, A1, A2, 15));
        cur.b0 = interpolate_attr(B, A0, A1, A2, 16); cur.b1 = interpolate_attr(B, A0, A1, A2, 17);
        cur.b2 = __uint_as_float(ob.material);
    }
    const auto second = [&](const float4 *, const float4 *, const float4 *, float4 &gc, float4 &gd, float4 &ge) {
        // attribute order (XVert::attr): uv 0-1, t 2-4, b 5-7, n 8-10, world 11-13, light space 14-17; planes as gbuffer_pack
        const auto mix = [&](float x0, float x1, float x2) {
#pragma clang fp contract(off)
            return (B[0] * x0 + B[1] * x1) + B[2] * x2;
        };
        if (compact) {
            const uint32_t a0 = v0 * 96u + 24u, a1 = v1 * 96u + 24u, a2 = v2 * 96u + 24u;   // attr 2..13: three float4 per vertex
            const float4u p0 = gload_f4u(sp.xv, a0), p1 = gload_f4u(sp.xv, a1), p2 = gload_f4u(sp.xv, a2);                     // attr 2..5
            const float4u q0 = gload_f4u(sp.xv, a0 + 16u), q1 = gload_f4u(sp.xv, a1 + 16u), q2 = gload_f4u(sp.xv, a2 + 16u);   // attr 6..9
            const float4u r0 = gload_f4u(sp.xv, a0 + 32u), r1 = gload_f4u(sp.xv, a1 + 32u), r2 = gload_f4u(sp.xv, a2 + 32u);   // attr 10..13
            gc = make_float4(mix(r0.y, r1.y, r2.y), mix(r0.z, r1.z, r2.z), mix(r0.w, r1.w, r2.w), mix(p0.x, p1.x, p2.x));
            gd = make_float4(mix(p0.y, p1.y, p2.y), mix(p0.z, p1.z, p2.z), mix(p0.w, p1.w, p2.w), mix(q0.x, q1.x, q2.x));
            ge = make_float4(mix(q0.y, q1.y, q2.y), mix(q0.z, q1.z, q2.z), mix(q0.w, q1.w, q2.w), mix(r0.x, r1.x, r2.x));
        } else {
            const float *A0 = sp.xv[v0].attr, *A1 = sp.xv[v1].attr, *A2 = sp.xv[v2].attr;
            gc = make_float4(interpolate_attr(B, A0, A1, A2, 11), interpolate_attr(B, A0, A1, A2, 12), interpolate_attr(B, A0, A1, A2, 13), interpolate_attr(B, A0, A1, A2, 2));
            gd = make_float4(interpolate_attr(B, A0, A1, A2, 3), interpolate_attr(B, A0, A1, A2, 4), interpolate_attr(B, A0, A1, A2, 5), interpolate_attr(B, A0, A1, A2, 6));
            ge = make_float4(interpolate_attr(B, A0, A1, A2, 7), interpolate_attr(B, A0, A1, A2, 8), interpolate_attr(B, A0, A1, A2, 9), interpolate_attr(B, A0, A1, A2, 10));
        }
    };
    const bool fast = shade_tile_fast<LOOP, STATS, false>(sp, args, A, lut, ty, tx, lane, cur, second);
    if (!fast) shade_tile<LOOP, STATS, LDS_SHADOW>(sp, lut, shadow_tiles[LDS_SHADOW ? wave : 0], ty, tx, lane, cur, second);
    trace_end(sp, A, (size_t)ty * A.tiles_x + tx, fast);
    if (++k >= A.T) break;
    asm volatile("" : "+s"(args));
    A = args_a(args);
    O = order_args(args);
    if (!next_tile(A, O, blk, k, tx, ty)) break;
    key = args->vis[((size_t)ty * A.tiles_x + tx) * 64 + wave_lane()];
    }
}

// ---- shadow bounds: the conservative min/max table calculate_lit tests first -------------------------------------------
// Two launches whenever the shadow map changes (64 MB read once at S = 4000; ~0.02 ms): blocks[j][i] = min/max of the 4x4
// texel block (i, j) (clamped at the map's edge), then entry (i, j) = blocks (i..i+1, j..j+1): texels [4i, 4i+8) x [4j, 4j+8).
__global__ __launch_bounds__(256) void k_shadow_blocks(const float *__restrict__ map, uint32_t S, uint32_t nb, float2 *__restrict__ blocks) {
    const uint32_t i = blockIdx.x * 64 + (threadIdx.x & 63), j = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (i >= nb || j >= nb) return;
    float lo = __builtin_inff(), hi = -__builtin_inff();
    for (uint32_t r = 0; r < 4; ++r) {
        const uint32_t y = 4 * j + r;
        if (y >= S) break;
        const float *row = map + (size_t)y * S;
        if (4 * i + 3 < S && (S & 3u) == 0u) {
            const float4 t = *reinterpret_cast<const float4 *>(row + 4 * i);
            lo = fminf(lo, fminf(fminf(t.x, t.y), fminf(t.z, t.w))); hi = fmaxf(hi, fmaxf(fmaxf(t.x, t.y), fmaxf(t.z, t.w)));
        } else {
            for (uint32_t c = 4 * i; c < min(4 * i + 4, S); ++c) { lo = fminf(lo, row[c]); hi = fmaxf(hi, row[c]); }
        }
    }
    blocks[(size_t)j * nb + i] = make_float2(lo, hi);
}
__global__ __launch_bounds__(256) void k_shadow_bounds(const float2 *__restrict__ blocks, uint32_t nb, float2 *__restrict__ bounds) {
    const uint32_t i = blockIdx.x * 64 + (threadIdx.x & 63), j = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (i >= nb || j >= nb) return;
    const uint32_t i1 = min(i + 1, nb - 1), j1 = min(j + 1, nb - 1);
    const float2 a = blocks[(size_t)j * nb + i], b = blocks[(size_t)j * nb + i1], c = blocks[(size_t)j1 * nb + i], d = blocks[(size_t)j1 * nb + i1];
    bounds[(size_t)j * nb + i] = make_float2(fminf(fminf(a.x, b.x), fminf(c.x, d.x)), fmaxf(fmaxf(a.y, b.y), fmaxf(c.y, d.y)));
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

template <int LOOP, bool STATS, bool LDS_SHADOW>
hipError_t launch_variant(const ShadeParams &sp, const ShadeLaunch &L, dim3 grid) {
    if (L.from_vis) {
        if (VIS_WG_WAVES == 1) grid.x = (grid.x + 7) / 8 * 32;   // (block_id, as below)
        k_material_vis<LOOP, STATS, LDS_SHADOW><<<grid, 64 * VIS_WG_WAVES, 0, L.stream>>>(sp);
        return hipGetLastError();
    }
    if (ARCTIC_WG_WAVES == 1) grid.x = (grid.x + 7) / 8 * 32;   // (block_id: four one-wave blocks per strip, a strip's blocks on one XCD)
    k_material<LOOP, STATS, LDS_SHADOW><<<grid, 64 * ARCTIC_WG_WAVES, 0, L.stream>>>(sp);
    return hipGetLastError();
}

}  // namespace

// The shading pass: one launch, one workgroup per 4 horizontally adjacent tiles (grid padded to whole groups of 8 tile rows
// for the XCD-aware order).  L.loop: 1 scalar light loop, 2 packed pairs.
hipError_t launch_shade(const ShadeParams &sp_in, const ShadeLaunch &L) {
    const uint32_t n_tiles = sp_in.tiles_x * sp_in.tiles_y;
    if (n_tiles == 0) return hipSuccess;
    ShadeParams sp = sp_in;
    // two tiles per wave pay once a frame is several rounds of the chip's wave slots; below ~3 Mpx one tile per wave is faster
    // (tools/experiments/small_frames.py: 1080p pass 0.0638 -> 0.0610 ms, whole frame 0.098 -> 0.084 ms with T = 1; 2688 x 1512: 0.101 against
    // 0.104 ms and 0.142 against 0.153 with T = 2)
    sp.tiles_per_wave = L.tiles_per_wave ? L.tiles_per_wave : (n_tiles < SMALL_FRAME_TILES ? 1u : DEFAULT_TILES_PER_WAVE);
    const uint32_t bpr = (sp.tiles_x + 3) / 4, groups = (sp.tiles_y + 7) / 8;
    sp.group_stride = (groups + sp.tiles_per_wave - 1) / sp.tiles_per_wave;
    dim3 grid(8 * bpr, sp.group_stride);   // a block shades tiles_per_wave groups of 8 tile rows, group_stride groups apart
    if (sp.tile_order)                     // ... or tiles_per_wave consecutive slots of the prepass's dispatch order (n_jobs slots, common.h order_slot: a block = one XCD's list)
        grid = dim3((sp.n_jobs + sp.tiles_per_wave - 1) / sp.tiles_per_wave, 1);
    if (sp.debug & 16)   // A/B only: the 25-tap path staged through LDS (a separate instantiation: it costs the default kernels nothing)
        return L.loop == 2 ? launch_variant<2, false, true>(sp, L, grid) : launch_variant<1, false, true>(sp, L, grid);
    if (L.loop == 2) return L.stats ? launch_variant<2, true, false>(sp, L, grid) : launch_variant<2, false, false>(sp, L, grid);
    return L.stats ? launch_variant<1, true, false>(sp, L, grid) : launch_variant<1, false, false>(sp, L, grid);
}

hipError_t launch_shadow_bounds(const float *map, uint32_t S, float2 *blocks, float2 *bounds, hipStream_t s) {
    const uint32_t nb = shadow_bounds_pitch(S);
    if (nb == 0) return hipSuccess;
    const dim3 grid((nb + 63) / 64, (nb + 3) / 4);
    k_shadow_blocks<<<grid, 256, 0, s>>>(map, S, nb, blocks);
    k_shadow_bounds<<<grid, 256, 0, s>>>(blocks, nb, bounds);
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
