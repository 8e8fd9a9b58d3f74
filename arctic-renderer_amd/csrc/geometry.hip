// geometry.hip -- visibility / G-buffer prepass and shadow-map raster for gfx950.
//
// Replaces what D3D12's fixed-function pipeline does for the reference around
// shaders/forward.hlsl:50-66 (vs_main) and shaders/depth.hlsl:7-10: vertex
// transform, clipping, back/front-face culling, rasterisation with the D3D
// rules (pixel centres at +0.5, 8 sub-pixel bits, top-left fill rule, depth
// LESS with first-drawn-wins ties: forward_pass.cpp:137-151,
// shadow_map_pass.cpp:96-97) and perspective-correct interpolation of the 18
// VSOut floats -- written out as the G-buffer instead of feeding a pixel shader.
//
// MI355X design: one launch per stage over ALL objects (no per-draw loop), no host read-back, no copies between the stages;
//   vertex  : 1 thread / vertex -> XVert[] (coalesced 96 B records); the launch also clears the pass's target and the counters
//   setup   : 1 thread / triangle inside all clip planes: set up, take record and work-item slots from one device counter (one
//             64-bit atomicAdd per workgroup) -> SetupRec[] + RasterRec[] in arbitrary order + explicit work items
//             (record, 16x16 block), only blocks an edge function can reach; triangles a plane cuts -> clip list.
//             Shadow pass (k_setup<true>, round 5): a triangle whose bounding box holds at most SMALL_PX pixels -- two thirds of what an
//             orthographic sun sees -- is drawn HERE, a lane per box pixel with 32-bit edge functions, and never becomes a record or a work item
//   clipped : the clip list, a lane per triangle: Sutherland-Hodgman in LDS, then the same set-up per fan triangle
//   bin + owned raster (forward pass): every 16x16 block of the target has a bin of 32 record indices and ONE owner wave that merges
//             the bin in registers and stores the block once -- no clear, no early read, no per-pixel atomic (k_bin, k_raster_owned)
//   raster  : (shadow pass; small shards; what the bins leave) persistent; edge functions as exact binary64 planes of the record, four pixels per lane; a wave sorts a chunk of
//             work items by block and merges each run in registers before ONE early read + atomicMin per pixel and run of
//             (depth bits << 32 | order id) into a tile-major visibility plane: order-independent, so no global sorting and
//             no per-pixel locks; order id = 8 * source triangle + sub-triangle keeps "first drawn wins" deterministic
//   resolve : 1 lane / pixel: fetch the winner, re-derive its barycentrics from the same planes, interpolate, write the
//             tile-major G-buffer planes (1 KiB / wave-store) (whole frames skip it: shade.hip's k_material_vis interpolates in place)
// The shadow map uses the same stages with a 32-bit atomicMin on the depth bits.
//
// This translation unit is compiled with -ffp-contract=off: coverage is integer and every fp32
// operation rounds once, in a fixed order, so visibility and G-buffer are bit-reproducible
// (the parity tests require bit equality with the CPU oracle).  fmaf() only where written.
#include "common.h"
#include "edges.h"
#include "shadow_coords.h"

namespace arctic {

namespace {

// Wave priority of the prepass kernels (s_setprio; 0 = the hardware's default).  A frame in flight runs its prepass beside the previous frame's shading
// kernel, whose seven waves per SIMD take the issue slots oldest first: a prepass kernel is a short dependent chain per workgroup, and every instruction
// of it waits its turn.  With a priority above the shading waves' the chain runs at its own length; the shading kernel gets the slots the chain leaves.
// Measured (profiles/r5_p_prepass_priority_ab.txt, libraries alternating on one box): whole 4K frames within noise (0.2645-0.2665 / 0.3465-0.3499 ->
// 0.2654-0.2658 / 0.3442-0.3474 ms), one rank of 8 as a row range 0.070 -> 0.066 ms, with the shadow map redrawn 0.164 -> 0.156; 3 measures like 1.
#ifndef ARCTIC_PREPASS_PRIO
#define ARCTIC_PREPASS_PRIO 1
#endif
#ifndef ARCTIC_RASTER_I32
#define ARCTIC_RASTER_I32 0   // 1: shadow-pass records whose edge functions fit 32 bits over their blocks are evaluated in integers (item_pixels).  Same map, 14 of 36 binary64-rate
                              // instructions per item -- and SLOWER like every trim of this kernel before it (k_raster<true> 50.7 -> 53.0 us, 71.6 -> 78.3 without the small path:
                              // profiles/r5_t_raster_int32_ab.txt): off
#endif
#ifndef ARCTIC_RASTER_WGW
#define ARCTIC_RASTER_WGW 4   // waves per workgroup of k_bin / k_raster_owned (their waves are independent of each other): an A/B switch beside ARCTIC_VIS_WG_WAVES (shade.hip)
#endif
__device__ __forceinline__ void prepass_priority() { if (ARCTIC_PREPASS_PRIO) __builtin_amdgcn_s_setprio(ARCTIC_PREPASS_PRIO); }

constexpr float GUARD = 64.0f;  // guard band |x|,|y| <= GUARD*w keeps 24.8 coordinates inside int32
constexpr int MAX_POLY = 10;

struct CV {  // clip-space vertex carried through the clipper
    float x, y, z, w;
    float b0, b1, b2;  // barycentrics w.r.t. the source triangle
};

__device__ __forceinline__ float plane_dist(const CV &v, int plane) {
    switch (plane) {
    case 0: return v.z;
    case 1: return v.w - v.z;
    case 2: return v.x + GUARD * v.w;
    case 3: return GUARD * v.w - v.x;
    case 4: return v.y + GUARD * v.w;
    default: return GUARD * v.w - v.y;
    }
}

// intersection from the inside vertex towards the outside vertex
__device__ __forceinline__ CV clip_lerp(const CV &in, const CV &out, float din, float dout) {
    float t = din / (din - dout);
    CV r;
    r.x = fmaf(t, out.x - in.x, in.x);
    r.y = fmaf(t, out.y - in.y, in.y);
    r.z = fmaf(t, out.z - in.z, in.z);
    r.w = fmaf(t, out.w - in.w, in.w);
    r.b0 = fmaf(t, out.b0 - in.b0, in.b0);
    r.b1 = fmaf(t, out.b1 - in.b1, in.b1);
    r.b2 = fmaf(t, out.b2 - in.b2, in.b2);
    return r;
}

// Sutherland-Hodgman against near, far and the four guard-band planes.  The polygons (up to MAX_POLY vertices, two buffers that
// change roles per plane) live in LDS, one column per lane: indexed private arrays would be scratch memory, ten times the latency.
constexpr uint32_t CLIP_LANES = 8;   // triangles per wave of k_setup_clipped: their large records are written one after the other by the whole wave
struct PolyStore {
    float v[2][MAX_POLY][7][CLIP_LANES];
    __device__ __forceinline__ CV get(int buf, int i, uint32_t lane) const {
        CV r; const float(*p)[CLIP_LANES] = v[buf][i];
        r.x = p[0][lane]; r.y = p[1][lane]; r.z = p[2][lane]; r.w = p[3][lane]; r.b0 = p[4][lane]; r.b1 = p[5][lane]; r.b2 = p[6][lane];
        return r;
    }
    __device__ __forceinline__ void put(int buf, int i, uint32_t lane, const CV &c) {
        float(*p)[CLIP_LANES] = v[buf][i];
        p[0][lane] = c.x; p[1][lane] = c.y; p[2][lane] = c.z; p[3][lane] = c.w; p[4][lane] = c.b0; p[5][lane] = c.b1; p[6][lane] = c.b2;
    }
};
// the triangle is in buffer 0, vertices 0..2; returns the vertex count of the clipped polygon and the buffer that holds it
__device__ __forceinline__ int clip_polygon(PolyStore &P, uint32_t lane, int &buf) {
    int n = 3;
    buf = 0;
    for (int plane = 0; plane < 6; ++plane) {
        bool all_in = true, any_in = false;
        for (int i = 0; i < n; ++i) {
            if (plane_dist(P.get(buf, i, lane), plane) >= 0.0f) any_in = true; else all_in = false;
        }
        if (all_in) continue;
        if (!any_in) return 0;
        int m = 0;
        for (int i = 0; i < n; ++i) {
            const int j = (i + 1 == n) ? 0 : i + 1;
            const CV vi = P.get(buf, i, lane), vj = P.get(buf, j, lane);
            const float di = plane_dist(vi, plane), dj = plane_dist(vj, plane);
            const bool in_i = di >= 0.0f, in_j = dj >= 0.0f;
            if (in_i) P.put(buf ^ 1, m++, lane, vi);
            if (in_i != in_j) P.put(buf ^ 1, m++, lane, in_i ? clip_lerp(vi, vj, di, dj) : clip_lerp(vj, vi, dj, di));
        }
        n = m;
        buf ^= 1;
        if (n < 3) return 0;
    }
    return n;
}

__device__ __forceinline__ int32_t snap(float s) { return (int32_t)floorf(s * 256.0f + 0.5f); }

// viewport transform, snapping, culling, orientation, pixel bounds.  false = nothing to rasterise.
__device__ __forceinline__ bool setup_triangle(const CV &a, const CV &b, const CV &c, const GeomParams &gp, SetupRec &t) {
    if (!(a.w > 0.0f) || !(b.w > 0.0f) || !(c.w > 0.0f)) return false;
    const float hx = 0.5f * gp.vp_w, hy = 0.5f * gp.vp_h;
    int32_t X[3], Y[3];
    float z[3], iw[3];
    const auto project = [&](const CV &v, int i) {
        iw[i] = 1.0f / v.w;
        const float nx = v.x * iw[i], ny = v.y * iw[i];
        z[i] = v.z * iw[i];
        X[i] = snap((nx + 1.0f) * hx);   // D3D viewport: X = (x+1) * W/2
        Y[i] = snap((1.0f - ny) * hy);   //               Y = (1-y) * H/2
    };
    project(a, 0); project(b, 1); project(c, 2);
    int64_t area2 = (int64_t)(X[1] - X[0]) * (int64_t)(Y[2] - Y[0]) - (int64_t)(X[2] - X[0]) * (int64_t)(Y[1] - Y[0]);
    if (area2 == 0) return false;
    // y-down: area2 > 0 <=> clockwise as seen; front = counter-clockwise (FrontCounterClockwise = TRUE)
    bool front = area2 < 0;
    if (gp.cull_front ? front : !front) return false;
    // oriented so that area2 > 0: vertices 1 and 2 change places (selects, no indexed private arrays: those live in scratch memory)
    const bool swap = area2 < 0;
    if (swap) area2 = -area2;
    t.X[0] = X[0]; t.Y[0] = Y[0]; t.z[0] = z[0]; t.iw[0] = iw[0];
    t.X[1] = swap ? X[2] : X[1]; t.Y[1] = swap ? Y[2] : Y[1]; t.z[1] = swap ? z[2] : z[1]; t.iw[1] = swap ? iw[2] : iw[1];
    t.X[2] = swap ? X[1] : X[2]; t.Y[2] = swap ? Y[1] : Y[2]; t.z[2] = swap ? z[1] : z[2]; t.iw[2] = swap ? iw[1] : iw[2];
    t.bary[0][0] = a.b0; t.bary[0][1] = a.b1; t.bary[0][2] = a.b2;
    t.bary[1][0] = swap ? c.b0 : b.b0; t.bary[1][1] = swap ? c.b1 : b.b1; t.bary[1][2] = swap ? c.b2 : b.b2;
    t.bary[2][0] = swap ? b.b0 : c.b0; t.bary[2][1] = swap ? b.b1 : c.b1; t.bary[2][2] = swap ? b.b2 : c.b2;
    t.area2 = area2;
    int32_t xmin = min(X[0], min(X[1], X[2])), xmax = max(X[0], max(X[1], X[2]));
    int32_t ymin = min(Y[0], min(Y[1], Y[2])), ymax = max(Y[0], max(Y[1], Y[2]));
    t.px0 = max((xmin - 128 + 255) >> 8, gp.sc_x0);
    t.px1 = min((xmax - 128) >> 8, gp.sc_x1 - 1);
    t.py0 = max((ymin - 128 + 255) >> 8, gp.sc_y0);
    t.py1 = min((ymax - 128) >> 8, gp.sc_y1 - 1);
    return t.px0 <= t.px1 && t.py0 <= t.py1;
}

// raster work items are 16x16-pixel blocks (one wave): the per-item record fetch is latency, so fewer, fatter items
__device__ __forceinline__ uint32_t tiles_of(const SetupRec &t) {
    return (uint32_t)((t.px1 >> 4) - (t.px0 >> 4) + 1) * (uint32_t)((t.py1 >> 4) - (t.py0 >> 4) + 1);
}

// the record's edge functions as binary64 planes over pixel coordinates (RasterRec, common.h).  edge_eval(e, i, px, py) =
// dx*(256 py + 128 - y0) - dy*(256 px + 128 - x0) = C + A px + B py with A = -256 dy, B = 256 dx, C = dx (128 - y0) - dy (128 - x0).
// With M = max |X|, |Y| < 2^24: |dx|, |dy| < 2^25, |128 - y0| < 2^25, so |C| < 2^51, |A px|, |B py| < 2^47 (px, py < 2^14):
// every product and partial sum is an integer below 2^53, i.e. exact in binary64 in any order.
__device__ __forceinline__ void make_raster_rec(const SetupRec &t, bool force_integer, RasterRec &q) {
    Edges e;
    make_edges(t, e);
    int32_t m = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        m = max(m, max(abs(t.X[i]), abs(t.Y[i])));
        const int64_t c = e.dx[i] * (int64_t)(128 - e.y0[i]) - e.dy[i] * (int64_t)(128 - e.x0[i]);   // 64-bit integers: exact whatever the magnitudes
        const int64_t thr = -e.bias[i];                                                               // covered <=> e + bias >= 0 <=> e >= -bias
        q.A[i] = (double)(-e.dy[i] * 256);
        q.B[i] = (double)(e.dx[i] * 256);
        q.C[i] = (double)(i == 1 ? c - thr : c);
        if (i == 0) q.t0 = (double)thr;
        if (i == 2) q.t2 = (double)thr;
    }
    q.z0 = t.z[0]; q.dz1 = t.z[1] - t.z[0]; q.dz2 = t.z[2] - t.z[0];
    q.inv_area = 1.0f / (float)t.area2;
    q.order_id = t.order_id;
    q.flags = (m < (1 << 24) && !force_integer) ? RASTER_EXACT_F64 : 0u;
    q.src_vertex[0] = q.src_vertex[1] = q.src_vertex[2] = 0; q.material = 0;   // (place_triangle)
}

// can any pixel of the 16x16 block (bx, by) be covered?  The largest value of each edge function over the block's pixel centres
// (at the corner its gradient points to) against the edge's threshold: exact, so a block is dropped only when it is empty.
__device__ __forceinline__ bool block_reachable(const RasterRec &q, int32_t bx, int32_t by) {
    const double x0 = (double)(bx * 16), y0 = (double)(by * 16);
    bool any = true;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double x = q.A[i] > 0.0 ? x0 + 15.0 : x0, y = q.B[i] > 0.0 ? y0 + 15.0 : y0;
        const double best = __builtin_fma(q.A[i], x, __builtin_fma(q.B[i], y, q.C[i]));
        const double thr = i == 0 ? q.t0 : (i == 2 ? q.t2 : 0.0);
        any = any && best >= thr;
    }
    return any;
}

// HLSL mul(M, v) with M column-major: row i = ((m0i*x + m1i*y) + m2i*z) + m3i*w
__device__ __forceinline__ void mat_vec(const float *m, float x, float y, float z, float w, float *o) {
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = ((m[i] * x + m[4 + i] * y) + m[8 + i] * z) + m[12 + i] * w;
}
__device__ __forceinline__ void normalize3(const float *v, float *o) {
    float inv = 1.0f / sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
    o[0] = v[0] * inv; o[1] = v[1] * inv; o[2] = v[2] * inv;
}

// Cluster culling (round 5; no counterpart in the reference, which hands every triangle to the hardware's own clipper and culler,
// forward_pass.cpp:212-224).  bb = {min xyz, max xyz} in object space of the positions a workgroup's triangles use (renderer.cpp:
// cluster_bounds).  True only when NO triangle with its vertices in the box can touch a pixel this pass keeps: all eight corners, transformed
// with the products k_vertex uses, lie beyond ONE plane -- near, far, or a side of the scissor rectangle `strict` pixels further out -- by more
// than the rounding those two products can accumulate (E x the sum of the magnitudes of their terms; the products' own bound is ~5e-7 of
// it).  The planes are half-spaces of clip space (x - x_l w < 0 ...), so no corner needs w > 0: whatever part of a triangle is drawn has
// w > 0 (setup_triangle, the clipper's near plane) and lies in the same half-space, i.e. projects outside the scissor.  A lane takes corner
// lane & 7; every wave of the workgroup reaches the same answer from the same numbers.  NaN or infinite bounds (renderer.cpp stores
// them for clusters it cannot bound) compare false everywhere: never culled.
// strict = 2 for k_vertex's vertex blocks, whose boxes contain the boxes of every cluster that uses one of their vertices: a block is
// skipped only if each of those clusters is skipped by k_setup's test with strict = 1 (twice the margins on a larger box).
__device__ __forceinline__ bool box_outside(const float *__restrict__ bb, const float *__restrict__ trs, const GeomParams &gp, float strict) {
    const uint32_t k = threadIdx.x & 7u;
    const float lo0 = bb[0], lo1 = bb[1], lo2 = bb[2], hi0 = bb[3], hi1 = bb[4], hi2 = bb[5];
    float world[4], clip[4], aw[4], ac[4];
    mat_vec(trs, (k & 1u) ? hi0 : lo0, (k & 2u) ? hi1 : lo1, (k & 4u) ? hi2 : lo2, 1.0f, world);
    mat_vec(gp.clip_from_world, world[0], world[1], world[2], world[3], clip);
    const float ax = fmaxf(fabsf(lo0), fabsf(hi0)), ay = fmaxf(fabsf(lo1), fabsf(hi1)), az = fmaxf(fabsf(lo2), fabsf(hi2));
#pragma unroll
    for (int i = 0; i < 4; ++i) aw[i] = ((fabsf(trs[i]) * ax + fabsf(trs[4 + i]) * ay) + fabsf(trs[8 + i]) * az) + fabsf(trs[12 + i]);
    const float *c = gp.clip_from_world;
#pragma unroll
    for (int i = 0; i < 4; ++i) ac[i] = ((fabsf(c[i]) * aw[0] + fabsf(c[4 + i]) * aw[1]) + fabsf(c[8 + i]) * aw[2]) + fabsf(c[12 + i]) * aw[3];
    const float E = 4.0e-6f * strict, ex = E * ac[0], ey = E * ac[1], ez = E * ac[2], ew = E * ac[3];
    const float hx = 0.5f * gp.vp_w, hy = 0.5f * gp.vp_h;   // setup_triangle: X = (x / w + 1) hx, Y = (1 - y / w) hy
    const float xl = ((float)gp.sc_x0 - strict) / hx - 1.0f, xr = ((float)gp.sc_x1 + strict) / hx - 1.0f;
    const float yt = 1.0f - ((float)gp.sc_y0 - strict) / hy, yb = 1.0f - ((float)gp.sc_y1 + strict) / hy;
    const bool in_near = !(clip[2] < -ez), in_far = !(clip[2] - clip[3] > ez + ew);
    const bool in_left = !(clip[0] - xl * clip[3] < -(ex + fabsf(xl) * ew)), in_right = !(clip[0] - xr * clip[3] > ex + fabsf(xr) * ew);
    const bool in_top = !(clip[1] - yt * clip[3] > ey + fabsf(yt) * ew), in_bottom = !(clip[1] - yb * clip[3] < -(ey + fabsf(yb) * ew));
    return !__ballot(in_near) || !__ballot(in_far) || !__ballot(in_left) || !__ballot(in_right) || !__ballot(in_top) || !__ballot(in_bottom);
}

// ---------------------------------------------------------------------------------------------
// forward.hlsl:50-66 vs_main for every vertex of every object in one launch
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_vertex(const ObjectRec *__restrict__ objs, const uint32_t *__restrict__ block_obj,
                                                const uint32_t *__restrict__ block_first, const GeomParams gp,
                                                XVert *__restrict__ xv, int clip_only, uint32_t *__restrict__ counters,
                                                unsigned long long *__restrict__ clear, unsigned long long clear_value, size_t clear_count,
                                                uint32_t *__restrict__ zero, size_t zero_count, const float *__restrict__ bounds) {
    prepass_priority();
    // k_setup's slot counters and overflow flag (no memset launch); [6] counts this kernel's own skipped blocks when asked to (raster_flags bit 1: zeroed by the host then)
    if (blockIdx.x == 0 && threadIdx.x < N_GEO_COUNTERS && !((gp.raster_flags & 2) && threadIdx.x == 6)) counters[threadIdx.x] = 0;
    const ObjectRec &ob = objs[block_obj[blockIdx.x]];
    uint32_t vi = block_first[blockIdx.x] + threadIdx.x;
    // vertices only triangles of culled clusters use are not transformed: nothing will read them (box_outside)
    const bool outside = bounds && box_outside(bounds + 6u * blockIdx.x, ob.trs, gp, 2.0f);   // (by every lane: the corners are spread over them)
    const bool live = vi < ob.n_vertices && !outside;
    if (outside && (gp.raster_flags & 2) && threadIdx.x == 0) atomicAdd(&counters[6], 1u);
    float src[14];
    if (live) {
        const float *p = ob.vertices + (size_t)vi * 14;
#pragma unroll
        for (int k = 0; k < 3; ++k) src[k] = p[k];
        if (!clip_only) {
#pragma unroll
            for (int k = 3; k < 14; ++k) src[k] = p[k];
        }
    }
    // the pass's target is cleared here, a slice per workgroup: 66 MB of stores (bandwidth) behind the vertex fetches just issued
    // (latency), instead of a launch of their own in front of them
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < clear_count; i += (size_t)gridDim.x * 256) clear[i] = clear_value;
    // ... or, when the blocks of the target have owners (k_raster_owned writes every pixel once), only the bins' counters
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < zero_count; i += (size_t)gridDim.x * 256) zero[i] = 0u;
    if (!live) return;
    float world[4];
    mat_vec(ob.trs, src[0], src[1], src[2], 1.0f, world);
    XVert &o = xv[ob.first_xvert + vi];
    float clip[4];
    mat_vec(gp.clip_from_world, world[0], world[1], world[2], world[3], clip);
    o.clip[0] = clip[0]; o.clip[1] = clip[1]; o.clip[2] = clip[2]; o.clip[3] = clip[3];
    if (clip_only) return;   // depth.hlsl:7-10
    float t[3], b[3], n[3], ls[4];
    normalize3(src + 6, t);
    normalize3(src + 3, n);
    normalize3(src + 9, b);
    mat_vec(gp.light_from_world, world[0], world[1], world[2], world[3], ls);
    o.attr[0] = src[12]; o.attr[1] = src[13];
    o.attr[2] = t[0]; o.attr[3] = t[1]; o.attr[4] = t[2];
    o.attr[5] = b[0]; o.attr[6] = b[1]; o.attr[7] = b[2];
    o.attr[8] = n[0]; o.attr[9] = n[1]; o.attr[10] = n[2];
    o.attr[11] = world[0]; o.attr[12] = world[1]; o.attr[13] = world[2];
    o.attr[14] = ls[0]; o.attr[15] = ls[1]; o.attr[16] = ls[2]; o.attr[17] = ls[3];
}

// ---------------------------------------------------------------------------------------------
// set-up + work-item expansion, one thread per source triangle (k_setup; cut triangles: k_setup_clipped).
// Record slots and work-item slots are taken from one 64-bit device counter with one atomicAdd per workgroup, so
// there is no count pass, no prefix scan and nothing for the host to wait for.  Records therefore land in arbitrary
// order; what must stay deterministic -- "first drawn wins" on equal depth -- travels in the record instead:
// order_id = 8 * (draw-order index of the source triangle) + (index of the sub-triangle the clipper produced), which is
// the low word of the visibility key.  rec_of[order_id] finds the record again in k_resolve.
// A work item is one (record, 16x16-pixel block) pair, stored explicitly: items[i] = {record, block coordinates + flags}
// (ITEM_SCISSOR, ITEM_SKIP: common.h).
// counters: [0] records, [1] work items, [2] set when a table overflowed (the frame is then incomplete), [3] clip-list entries.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t v, uint32_t lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(v, d); if (lane >= (uint32_t)d) v += y; }
    return v;
}

struct SetupTables {
    SetupRec *recs; RasterRec *rrecs; uint32_t *rec_of; uint2 *items; uint32_t item_cap, rec_cap;
    uint32_t *counters;   // [0] records, [1] work items (one 64-bit word), [2] overflow flag, [3] entries of the clip list
    uint32_t *depth_bits; // k_setup<true> (the shadow pass): the map, for the triangles k_setup draws itself (draw_small)
};

// clip-space vertices of source triangle ti of object ob; false: no such triangle / an index out of range
__device__ __forceinline__ bool load_indices(const ObjectRec &ob, uint32_t ti, uint32_t &i0, uint32_t &i1, uint32_t &i2) {
    i0 = i1 = i2 = 0u;
    if (ti >= ob.n_triangles) return false;
    i0 = ob.indices[3 * ti]; i1 = ob.indices[3 * ti + 1]; i2 = ob.indices[3 * ti + 2];
    return i0 < ob.n_vertices && i1 < ob.n_vertices && i2 < ob.n_vertices;
}
__device__ __forceinline__ bool load_triangle(const ObjectRec &ob, uint32_t ti, const XVert *__restrict__ xv, CV &a, CV &b, CV &c, const float *__restrict__ cull_bounds = nullptr,
                                              const GeomParams *gp = nullptr, bool *culled = nullptr) {
    uint32_t i0, i1, i2;
    const bool ok = load_indices(ob, ti, i0, i1, i2);
    // (k_setup: the cluster's test sits between the index loads and the vertex loads that wait for them, so a cluster that stays pays no round trip for it)
    if (cull_bounds && box_outside(cull_bounds, ob.trs, *gp, 1.0f)) { *culled = true; return false; }
    if (!ok) return false;
    const auto fetch = [&](uint32_t i, CV &v, float b0, float b1, float b2) {
        const float4 p = *reinterpret_cast<const float4 *>(xv[ob.first_xvert + i].clip);
        v.x = p.x; v.y = p.y; v.z = p.z; v.w = p.w; v.b0 = b0; v.b1 = b1; v.b2 = b2;
    };
    fetch(i0, a, 1.0f, 0.0f, 0.0f); fetch(i1, b, 0.0f, 1.0f, 0.0f); fetch(i2, c, 0.0f, 0.0f, 1.0f);
    return true;
}
typedef int32_t i2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int32_t high_word(double v) { return __builtin_bit_cast(i2v, v).y; }   // sign and exponent: negative <=> the word is

// records of up to LANE_BLOCKS blocks are expanded by their own lane (a 64-bit mask of reachable blocks); larger ones by the whole wave,
// one record after the other -- with the limit at 16 a wave of 64 mid-sized triangles (a wall at 5 m: 7 x 7 blocks each) spent 20 us there
constexpr uint32_t LANE_BLOCKS = 64;
struct EmitPlan {
    RasterRec q;
    int32_t bx0, by0;
    uint32_t nbx, nbb /*blocks of the bounding box*/, nb /*item slots*/;
    unsigned long long mask;   // reachable blocks of a record of up to 64
};
// does this shard own any pixel row of block row `by` (16 rows = tile rows 2 by, 2 by + 1)?  A contiguous shard is scissored to its
// rows already; an interleaved one owns bands of band_tiles tile rows: blocks wholly inside other shards' bands are never emitted
// (the rasteriser would walk them and skip every row: at 8 ranks 7 of 8 items).
__device__ __forceinline__ bool block_row_owned(const GeomParams &gp, int32_t by) {
    if (gp.band_tiles == 0) return true;
    const int t0 = 2 * by - gp.tile_y0;
    return (t0 >= 0 && row_owned(t0, gp.band_tiles, gp.shard_count, gp.shard_index)) || (t0 + 1 >= 0 && row_owned(t0 + 1, gp.band_tiles, gp.shard_count, gp.shard_index));
}
// (has is cleared when no block of the triangle is both reachable and owned: no record, no slot)
__device__ __forceinline__ void plan_triangle(bool &has, const SetupRec &t, const GeomParams &gp, EmitPlan &e) {
    e.q = RasterRec{};
    e.bx0 = e.by0 = 0; e.nbx = 1; e.nbb = e.nb = 0; e.mask = 0ull;
    if (!has) return;
    make_raster_rec(t, (gp.raster_flags & 1) != 0, e.q);
    e.bx0 = t.px0 >> 4; e.by0 = t.py0 >> 4;
    e.nbx = (uint32_t)((t.px1 >> 4) - e.bx0 + 1);
    e.nbb = tiles_of(t);
    e.nb = e.nbb;
#if ARCTIC_RASTER_I32
    if (gp.pitch != 0 && (e.q.flags & RASTER_EXACT_F64)) {   // (the shadow pass: gp.pitch is its map's)
        // a plane is largest in magnitude at a corner of the block-aligned bounding box
        const double xa = (double)(e.bx0 * 16), xb = (double)((t.px1 >> 4) * 16 + 15), ya = (double)(e.by0 * 16), yb = (double)((t.py1 >> 4) * 16 + 15);
        double m = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double ca = __builtin_fma(e.q.B[i], ya, e.q.C[i]), cb = __builtin_fma(e.q.B[i], yb, e.q.C[i]);
            m = fmax(m, fmax(fmax(fabs(__builtin_fma(e.q.A[i], xa, ca)), fabs(__builtin_fma(e.q.A[i], xb, ca))), fmax(fabs(__builtin_fma(e.q.A[i], xa, cb)), fabs(__builtin_fma(e.q.A[i], xb, cb)))));
            m = fmax(m, fmax(fabs(e.q.A[i]), fabs(e.q.B[i])) * 64.0);
        }
        if (m < 2147480000.0) e.q.flags |= RASTER_I32;
    }
#endif
    if (e.nbb <= LANE_BLOCKS && (e.q.flags & RASTER_EXACT_F64)) {
        // block_reachable for every block of the bounding box, incrementally: per edge the value at the first block's best corner,
        // then + 16 A per block to the right, + 16 B per block down (exact: integers below 2^53); reachable <=> no negative value
        const RasterRec &q = e.q;
        double row[3], sx[3], sy[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double x = (double)(e.bx0 * 16) + (q.A[i] > 0.0 ? 15.0 : 0.0), y = (double)(e.by0 * 16) + (q.B[i] > 0.0 ? 15.0 : 0.0);
            row[i] = __builtin_fma(q.A[i], x, __builtin_fma(q.B[i], y, q.C[i])) - (i == 0 ? q.t0 : (i == 2 ? q.t2 : 0.0));
            sx[i] = 16.0 * q.A[i]; sy[i] = 16.0 * q.B[i];
        }
        double v[3] = {row[0], row[1], row[2]};
        uint32_t x = 0;
        int32_t y = e.by0;
        bool mine = block_row_owned(gp, y), any_mine = mine;
        for (uint32_t j = 0; j < e.nbb; ++j) {
            if (mine && (high_word(v[0]) | high_word(v[1]) | high_word(v[2])) >= 0) e.mask |= 1ull << j;
            if (++x == e.nbx) { x = 0; mine = block_row_owned(gp, ++y); any_mine |= mine && j + 1 < e.nbb; row[0] += sy[0]; row[1] += sy[1]; row[2] += sy[2]; v[0] = row[0]; v[1] = row[1]; v[2] = row[2]; }
            else { v[0] += sx[0]; v[1] += sx[1]; v[2] += sx[2]; }
        }
        e.nb = (uint32_t)__popcll(e.mask);
        if (!any_mine) { has = false; e.nbb = 0u; }   // (a triangle none of whose OWN blocks is reachable keeps its record: it counts as set up, like on one device)
    } else if (e.nbb <= LANE_BLOCKS) {
        uint32_t x = 0;
        int32_t y = e.by0;
        for (uint32_t j = 0; j < e.nbb; ++j) {
            if (block_row_owned(gp, y)) e.mask |= 1ull << j;
            if (++x == e.nbx) { x = 0; ++y; }
        }
        e.nb = (uint32_t)__popcll(e.mask);
        if (e.nb == 0u) { has = false; e.nbb = 0u; }
    }
}
// record slot r and item slots [ibase, ibase + e.nb) are this lane's: write them.  Called by whole waves (large records are
// written by all lanes together).
__device__ __forceinline__ void place_triangle(bool has, SetupRec &t, EmitPlan &e, uint32_t r, uint32_t ibase, uint32_t src, uint32_t oi, uint32_t sub,
                                               const GeomParams &gp, const SetupTables &T, const ObjectRec &ob) {
    const uint32_t lane = threadIdx.x & 63;
    if (has) {
        t.src_tri = src; t.object = oi; t.order_id = src * 8u + sub; t.pad = 0;
        e.q.order_id = t.order_id;
        // what shading from the visibility plane needs of the SOURCE triangle, so that it finds it in the record it reads anyway instead
        // of behind three more dependent loads (object -> index buffer -> ...): its transformed vertices and its material
        const uint32_t *ind = ob.indices + 3u * (src - ob.first_triangle);
        e.q.src_vertex[0] = ob.first_xvert + ind[0]; e.q.src_vertex[1] = ob.first_xvert + ind[1]; e.q.src_vertex[2] = ob.first_xvert + ind[2];
        e.q.material = ob.material;
        if (r < T.rec_cap) { T.recs[r] = t; T.rrecs[r] = e.q; T.rec_of[src * 8u + sub] = r; }
        else e.nb = e.nbb = 0;   // record table full (flagged by the caller; k_raster then does nothing)
    }
    const auto block_code = [&](int32_t bx, int32_t by, bool exact) {
        const bool whole = bx * 16 >= gp.sc_x0 && bx * 16 + 16 <= gp.sc_x1 && by * 16 >= gp.sc_y0 && by * 16 + 16 <= gp.sc_y1;
        return (uint32_t)bx | ((uint32_t)by << 12) | (whole ? 0u : ITEM_SCISSOR) | (exact ? 0u : ITEM_INEXACT);
    };
    // small records: the lane writes its own items; large ones (a wall across the screen is thousands of blocks): the
    // whole wave writes them, 64 per step
    if (e.nbb <= LANE_BLOCKS && e.nb != 0) {
        const bool own_exact = (e.q.flags & RASTER_EXACT_F64) != 0;
        uint32_t x = 0, y = 0, at = ibase;
        for (uint32_t j = 0; j < e.nbb; ++j) {
            if ((e.mask >> j) & 1ull) { if (at < T.item_cap) T.items[at] = make_uint2(r, block_code(e.bx0 + (int32_t)x, e.by0 + (int32_t)y, own_exact)); ++at; }
            if (++x == e.nbx) { x = 0; ++y; }
        }
    }
    for (unsigned long long big = __ballot(e.nbb > LANE_BLOCKS); big != 0ull; big &= big - 1ull) {
        const int L = __ffsll((long long)big) - 1;
        const uint32_t R = __shfl(r, L), NB = __shfl(e.nbb, L), IB = __shfl(ibase, L), NX = __shfl(e.nbx, L);
        const int32_t X0 = __shfl(e.bx0, L), Y0 = __shfl(e.by0, L);
        RasterRec w;
#pragma unroll
        for (int i = 0; i < 3; ++i) { w.A[i] = __shfl(e.q.A[i], L); w.B[i] = __shfl(e.q.B[i], L); w.C[i] = __shfl(e.q.C[i], L); }
        w.t0 = __shfl(e.q.t0, L); w.t2 = __shfl(e.q.t2, L);
        const bool exact = (__shfl(e.q.flags, L) & RASTER_EXACT_F64) != 0;
        for (uint32_t j = lane; j < NB; j += 64) {
            const int32_t bx = X0 + (int32_t)(j % NX), by = Y0 + (int32_t)(j / NX);
            const bool live = block_row_owned(gp, by) && (!exact || block_reachable(w, bx, by));
            if (IB + j < T.item_cap) T.items[IB + j] = make_uint2(live ? R : ITEM_SKIP, block_code(bx, by, exact));
        }
    }
}
// slots for nr records and ni work items: ONE 64-bit atomicAdd (records low, items high; same-address atomics retire at ~88
// per microsecond: one per wave was 40 us of k_setup at 4K)
__device__ __forceinline__ unsigned long long take_slots(uint32_t nr, uint32_t ni, const SetupTables &T) {
    const unsigned long long base = (nr | ni) ? atomicAdd(reinterpret_cast<unsigned long long *>(T.counters), (unsigned long long)nr | ((unsigned long long)ni << 32)) : 0ull;
    if ((uint32_t)(base >> 32) + ni > T.item_cap || (uint32_t)base + nr > T.rec_cap) T.counters[2] = 1;   // a table is full: reported by the host
    return base;
}

// ---------------------------------------------------------------------------------------------
// Small triangles of the shadow pass, drawn by k_setup itself (round 5).  The sun's orthographic frustum holds the whole scene, so most of
// what it sees is small: 63 % of config 3's 113 k shadow records have a bounding box of at most 64 pixels (median 54) -- and each paid
// a record (256 B written, 128 B read per item through the scalar cache), 1.8 work items, a place in a wave's sort and 256 pixel
// evaluations per item in binary64 for its ~25 covered pixels.  Here such a triangle never becomes a record: the lanes of the wave that
// set the triangles up share out the PIXELS of their bounding boxes (a lane per pixel, box after box, 64 at a time), the planes as 32-bit
// integers relative to the box's first pixel (exact: SmallRec), and a covered pixel goes to the map with one atomicMin -- lanes of one
// instruction that fall into one 64-byte row segment are one request for the memory side, whichever triangle they belong to, and the
// boxes of consecutive triangles of a mesh are neighbours.  Same coverage rule, same depth expression on the same exact numerators as
// item_pixels / raster_item_i64, and atomicMin does not care who sends the depth: the map is the same bit for bit.
// ---------------------------------------------------------------------------------------------
#ifndef SMALL_PX
#define SMALL_PX 64     // bounding-box pixels up to which a shadow-pass triangle is drawn by k_setup (0: none)
#endif
struct SmallRec {       // 64 B in LDS, one per small triangle of the wave, in lane order
    int32_t E[3];       // edge functions (edges.h: edge_eval + bias) at the box's first pixel: covered <=> all three >= 0
    int32_t Ax[3], By[3];   // their steps per pixel in x and y; |.| < 2^23 (24-bit multiplies), so that no value over the box leaves int32
    float z0, dz1, dz2, inv_area;
    uint32_t origin;    // px0 | py0 << 16
    uint32_t span;      // first pixel slot of the triangle in the wave's list | box width << 16 | (bias of edge 0 != 0) << 28 | (bias of edge 2 != 0) << 29
    uint32_t M;         // floor(65536 / width) + 1: slot j of the box is pixel (j - w (j M >> 16), j M >> 16) -- exact while w * j < 65536
};
static_assert(sizeof(SmallRec) == 64, "SmallRec layout");
constexpr uint32_t SMALL_WORDS = (SMALL_PX * 64u + 31u) / 32u + 2u;   // start bits of a wave's pixel list (+ the 64-bit read at its end)
static_assert(SMALL_PX <= 1023, "SmallRec::span holds a slot of the wave's list (64 SMALL_PX of them) in 16 bits, and slot -> box coordinates is exact while 64 * slot < 65536");

// is the triangle small, and if so its SmallRec (span without the slot, which the caller knows after the wave's prefix sum)
__device__ __forceinline__ bool small_record(const SetupRec &t, SmallRec &s, uint32_t &n) {
    const uint32_t w = (uint32_t)(t.px1 - t.px0 + 1), h = (uint32_t)(t.py1 - t.py0 + 1);
    n = w * h;
    if (w > 64u || h > (uint32_t)SMALL_PX || n > (uint32_t)SMALL_PX) return false;
    Edges e;
    make_edges(t, e);
    bool fits = true;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int64_t E = edge_eval(e, i, t.px0, t.py0) + e.bias[i], ax = -e.dy[i] * 256, by = e.dx[i] * 256;
        const int64_t reach = (E < 0 ? -E : E) + (ax < 0 ? -ax : ax) * (int64_t)(w - 1) + (by < 0 ? -by : by) * (int64_t)(h - 1);
        fits = fits && ax > -(1 << 23) && ax < (1 << 23) && by > -(1 << 23) && by < (1 << 23) && reach < 0x7FFFFFF0ll;
        s.E[i] = (int32_t)E; s.Ax[i] = (int32_t)ax; s.By[i] = (int32_t)by;
    }
    s.z0 = t.z[0]; s.dz1 = t.z[1] - t.z[0]; s.dz2 = t.z[2] - t.z[0];
    s.inv_area = 1.0f / (float)t.area2;
    s.origin = (uint32_t)t.px0 | ((uint32_t)t.py0 << 16);
    s.span = (w << 16) | (e.bias[0] != 0 ? 1u << 28 : 0u) | (e.bias[2] != 0 ? 1u << 29 : 0u);
    s.M = 65536u / w + 1u;
    return fits && t.px0 >= 0 && t.py0 >= 0 && t.px1 < 65536 && t.py1 < 65536;
}

// every lane of the wave calls this; `small` lanes bring a triangle.  recs / starts: this wave's LDS.  Returns the wave's pixel slots.
// (Two parts, so that the sixteen registers of a SmallRec do not live through planning and placing the other triangles.)
__device__ __forceinline__ uint32_t stage_small(bool small, SmallRec &s, uint32_t n, SmallRec *recs, uint32_t *starts) {
    const uint32_t lane = threadIdx.x & 63;
    const unsigned long long sm = __ballot(small);
    if (sm == 0ull) return 0u;   // uniform per wave
    const uint32_t incl = wave_inclusive_sum(small ? n : 0u, lane);
    for (uint32_t k = lane; k < SMALL_WORDS; k += 64) starts[k] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // (LDS operations of one wave execute in order; this keeps the compiler from exchanging them)
    if (small) {
        const uint32_t first = incl - n;
        s.span |= first;
        recs[__popcll(sm & ((1ull << lane) - 1ull))] = s;
        atomicOr(&starts[first >> 5], 1u << (first & 31u));
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    return __shfl(incl, 63);
}
__device__ __forceinline__ void draw_small(uint32_t total, uint32_t *__restrict__ depth_bits, uint32_t pitch, const SmallRec *recs, const uint32_t *starts) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t le_lo = lane >= 32 ? 0xFFFFFFFFu : (2u << lane) - 1u, le_hi = lane >= 32 ? (lane == 63 ? 0xFFFFFFFFu : (2u << (lane - 32)) - 1u) : 0u;   // lanes <= this one
    uint32_t base = 0;        // triangles that begin before this step's pixels
    for (uint32_t first = 0; first < total; first += 64) {
        const uint32_t w0 = starts[first >> 5], w1 = starts[(first >> 5) + 1];   // the step's 64 start bits (first is a multiple of 64)
        const uint32_t idx = base + (uint32_t)__popc(w0 & le_lo) + (uint32_t)__popc(w1 & le_hi) - 1u;
        base += (uint32_t)__popc(w0) + (uint32_t)__popc(w1);
        const uint32_t p = first + lane;
        if (p < total) {
            const SmallRec r = recs[idx];
            const uint32_t j = p - (r.span & 0xFFFFu), w = (r.span >> 16) & 0xFFFu;
            const uint32_t dy = __umul24(j, r.M) >> 16, dx = j - __umul24(dy, w);
            const int32_t e0 = r.E[0] + __mul24(r.Ax[0], (int32_t)dx) + __mul24(r.By[0], (int32_t)dy);
            const int32_t e1 = r.E[1] + __mul24(r.Ax[1], (int32_t)dx) + __mul24(r.By[1], (int32_t)dy);
            const int32_t e2 = r.E[2] + __mul24(r.Ax[2], (int32_t)dx) + __mul24(r.By[2], (int32_t)dy);
            // the depth's numerators are the edge functions without the fill-rule bias (bias = -1 where bit set)
            const float l1 = (float)(e2 + (int32_t)((r.span >> 29) & 1u)) * r.inv_area, l2 = (float)(e0 + (int32_t)((r.span >> 28) & 1u)) * r.inv_area;
            float z = fmaf(l2, r.dz2, fmaf(l1, r.dz1, r.z0));
            z = fminf(fmaxf(z, 0.0f), 1.0f);
            const uint32_t zb = __float_as_uint(z);
            if ((e0 | e1 | e2) >= 0 && zb < 0x3F800000u) {   // covered, and LESS against the 1.0 clear (a -0.0 never wins an unsigned min either way)
                const uint32_t at = __umul24((r.origin >> 16) + dy, pitch) + (r.origin & 0xFFFFu) + dx;
                __hip_atomic_fetch_min((uint32_t __attribute__((address_space(1))) *)((char __attribute__((address_space(1))) *)depth_bits + at * 4u), zb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// The common case, no private arrays and no LDS polygons: triangles inside all six planes are set up and emitted here (the
// workgroup takes its slots with one atomic, split among the waves through LDS); triangles a plane cuts go to the clip list
// for k_setup_clipped.  (With the clipper inside, its polygons in scratch memory, this kernel took 43 us at 4K.)
template <bool SMALL>   // SMALL: the shadow pass -- triangles with a small bounding box are drawn here (draw_small) and take no record slot's bytes, no work item
__global__ __launch_bounds__(SETUP_THREADS) void k_setup(const ObjectRec *__restrict__ objs, const uint32_t *__restrict__ block_obj,
                                                         const uint32_t *__restrict__ block_first, const GeomParams gp,
                                                         const XVert *__restrict__ xv, SetupTables T, uint2 *__restrict__ clip_list, const float *__restrict__ bounds) {
    prepass_priority();
    constexpr uint32_t WAVES = SETUP_THREADS / 64;
    __shared__ uint32_t s_count[WAVES][2], s_base[2];
    __shared__ SmallRec s_small[SMALL ? WAVES * 64 : 1];
    __shared__ uint32_t s_starts[SMALL ? WAVES * SMALL_WORDS : 1];
    const uint32_t oi = block_obj[blockIdx.x];
    const ObjectRec &ob = objs[oi];
    const uint32_t ti = block_first[blockIdx.x] + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    CV a, b, c;
    bool culled = false;
    const bool valid = load_triangle(ob, ti, xv, a, b, c, bounds ? bounds + 6u * blockIdx.x : nullptr, &gp, &culled);
    if (culled) {         // the whole workgroup: no record, no item, no entry of the clip list could have come from it
        if ((gp.raster_flags & 2) && threadIdx.x == 0) atomicAdd(&T.counters[5], 1u);
        return;
    }
    // the clipper's own first decision (clip_polygon): the first plane that does not hold all three vertices either holds none
    // -- nothing to draw -- or cuts the triangle; no such plane: the triangle goes through untouched
    bool inside = valid, straddles = false;
#pragma unroll
    for (int p = 5; p >= 0; --p) {
        const bool ia = plane_dist(a, p) >= 0.0f, ib = plane_dist(b, p) >= 0.0f, ic = plane_dist(c, p) >= 0.0f;
        if (!(ia && ib && ic)) { inside = false; straddles = valid && (ia || ib || ic); }   // descending p: the first such plane decides last
    }
    const unsigned long long cm = __ballot(straddles);
    if (cm) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&T.counters[3], (uint32_t)__popcll(cm));
        base = __shfl(base, 0);
        if (straddles) clip_list[base + (uint32_t)__popcll(cm & ((1ull << lane) - 1ull))] = make_uint2(oi, ti);
    }
    SetupRec t;
    bool has = inside && setup_triangle(a, b, c, gp, t);
    // (a small triangle counts as a record -- arctic_stats: triangles set up -- and takes a slot NUMBER, but nothing is written there and no item names it)
    SmallRec sr;
    uint32_t small_n = 0;
    const bool small = SMALL && has && small_record(t, sr, small_n);
    const uint32_t small_total = SMALL ? stage_small(small, sr, small_n, s_small + wave * 64, s_starts + wave * SMALL_WORDS) : 0u;
    const unsigned long long m = __ballot(has);
    has = has && !small;
    EmitPlan e;
    plan_triangle(has, t, gp, e);
    // an uncut source triangle: its vertices carry unit barycentrics (load_triangle), setup_triangle may have exchanged the last two
    if (has) e.q.flags |= RASTER_UNIT_BARY | (t.bary[1][2] == 1.0f ? RASTER_SWAPPED : 0u);
    const uint32_t iincl = wave_inclusive_sum(e.nb, lane);
    if (lane == 63) { s_count[wave][0] = (uint32_t)__popcll(m); s_count[wave][1] = iincl; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t nr = 0, ni = 0;
        for (uint32_t w = 0; w < WAVES; ++w) { nr += s_count[w][0]; ni += s_count[w][1]; }
        const unsigned long long base = take_slots(nr, ni, T);
        s_base[0] = (uint32_t)base; s_base[1] = (uint32_t)(base >> 32);
    }
    __syncthreads();
    uint32_t rbase = s_base[0], ibase = s_base[1];
    for (uint32_t w = 0; w < wave; ++w) { rbase += s_count[w][0]; ibase += s_count[w][1]; }
    place_triangle(has, t, e, rbase + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)), ibase + iincl - e.nb, ob.first_triangle + ti, oi, 0u, gp, T, ob);
    if (SMALL) draw_small(small_total, T.depth_bits, (uint32_t)gp.pitch, s_small + wave * 64, s_starts + wave * SMALL_WORDS);
}

// The triangles of the clip list, one wave per workgroup.  First a lane per triangle (CLIP_LANES of them): Sutherland-Hodgman in
// LDS.  Then a lane per FAN triangle -- 7 lanes per polygon (6 planes cut a triangle into at most 7), 8 polygons per round --, so
// set-up, planning, one slot atomic and the writes happen once per round instead of once (twice: count, then write) per fan index.
__global__ __launch_bounds__(64) void k_setup_clipped(const ObjectRec *__restrict__ objs, const GeomParams gp, const XVert *__restrict__ xv,
                                                      SetupTables T, const uint2 *__restrict__ clip_list) {
    prepass_priority();
    __shared__ PolyStore P;
    __shared__ int s_n[CLIP_LANES], s_buf[CLIP_LANES];
    __shared__ uint32_t s_obj[CLIP_LANES], s_src[CLIP_LANES];
    const uint32_t count = T.counters[3], lane = threadIdx.x;
    for (uint32_t first = blockIdx.x * CLIP_LANES; first < count; first += gridDim.x * CLIP_LANES) {   // uniform per wave
        if (lane < CLIP_LANES) {
            int n = 0, buf = 0;   // vertices of the clipped polygon (0: none) and the buffer that holds it
            uint32_t oi = 0, src = 0;
            if (first + lane < count) {
                const uint2 entry = clip_list[first + lane];
                oi = entry.x;
                const ObjectRec &ob = objs[oi];
                src = ob.first_triangle + entry.y;
                CV a, b, c;
                if (load_triangle(ob, entry.y, xv, a, b, c)) {
                    P.put(0, 0, lane, a); P.put(0, 1, lane, b); P.put(0, 2, lane, c);
                    n = clip_polygon(P, lane, buf);
                }
            }
            s_n[lane] = n; s_buf[lane] = buf; s_obj[lane] = oi; s_src[lane] = src;
        }
        __syncthreads();   // one wave: orders the LDS writes above before the reads below
        for (uint32_t round = 0; round < CLIP_LANES / 8; ++round) {
            const uint32_t p = round * 8 + lane / 7;   // lanes 56..63 idle
            const int f = (int)(lane % 7) + 1;
            SetupRec t;
            bool has = false;
            if (lane < 56 && f + 1 < s_n[p]) {
                const int buf = s_buf[p];
                has = setup_triangle(P.get(buf, 0, p), P.get(buf, f, p), P.get(buf, f + 1, p), gp, t);
            }
            EmitPlan e;
            plan_triangle(has, t, gp, e);
            const unsigned long long m = __ballot(has);
            const uint32_t iincl = wave_inclusive_sum(e.nb, lane);
            unsigned long long base = 0;
            if (lane == 63) base = take_slots((uint32_t)__popcll(m), iincl, T);
            base = __shfl(base, 63);
            const unsigned long long below = m & ((1ull << lane) - 1ull);
            const uint32_t sub = (uint32_t)__popcll(below >> ((lane / 7) * 7));   // set-up fan triangles of the same polygon before this one
            place_triangle(has, t, e, (uint32_t)base + (uint32_t)__popcll(below), (uint32_t)(base >> 32) + iincl - e.nb,
                           lane < 56 ? s_src[p] : 0u, lane < 56 ? s_obj[p] : 0u, sub, gp, T, objs[lane < 56 ? s_obj[p] : 0u]);
        }
        __syncthreads();   // the polygons are read; the next sweep may overwrite them
    }
}

// ---------------------------------------------------------------------------------------------
// raster: one wavefront per (triangle, 16x16 block) work item, four pixels per lane
// ---------------------------------------------------------------------------------------------
// what k_raster needs of GeomParams, read once per wave
struct RasterFrame {
    int32_t sc_x0, sc_y0, sc_x1, sc_y1;
    int32_t tiles_x, tile_y0, pitch, band_tiles, shard_index, shard_count;
};

// One work item through the binary64 planes of its RasterRec (exact, see make_raster_rec).  The record sits in scalar registers;
// nothing is set up per item: a lane evaluates the three planes at its own pixel coordinates, 15-18 v_fma_f64 for its four pixels.
// The pixels of a lane are chosen so that every wave-wide memory instruction touches the fewest lines: the forward pass takes the
// block as its four 8x8 tiles of the tile-major visibility plane (pixel k of lane l = pixel l of tile k: one 512-byte run per
// load / atomic), the shadow pass as four 16x4 strips of the row-major map (four 64-byte rows).  Covered <=> no sign bit among the
// three thresholded edge values (one v_or3 on the high words); the depth is edges.h's / the oracle's expression on the same exact
// numerators.
typedef char __attribute__((address_space(1))) *gbytes;   // wave-uniform base (SGPR pair) + 32-bit per-lane byte offset: the saddr form, no 64-bit address arithmetic

// the four pixels of a lane in one work item: the depth bits where the pixel is covered and nearer than the clear value, else NONE
constexpr uint32_t NO_DEPTH = 0xFFFFFFFFu;

// RAW (the block owners): the depth bits of every covered pixel, 0x3F800000 (1.0: fails LESS against the clear value) included --
// the owner sorts those out once per block instead of once per item --, and the fill-rule thresholds of edges 0 and 2 are taken
// off the high words as integers instead of off the values in binary64: for an integer-valued double e and t in {0, 1},
// e >= t  <=>  high word of e, as int32, >= t  (e = 0 has high word 0, e >= 1 a high word of 0x3FF00000 and more, e < 0 a negative
// one; the planes never produce -0.0: their coefficients come from integers and an exact zero sum rounds to +0.0).
template <bool DEPTH_ONLY, bool RAW = false>
__device__ __forceinline__ void item_pixels(const RasterRec &t, uint32_t code, uint32_t lane, const RasterFrame &fr, uint32_t zb[4]) {
    const int32_t ox = (int32_t)(code & 0xFFFu) << 4, oy = (int32_t)((code >> 12) & 0xFFFu) << 4;   // the block's first pixel
    const bool cut = (code & ITEM_SCISSOR) != 0;   // the block is cut by the scissor: test every pixel against it
    const int32_t x0 = ox + (int32_t)(DEPTH_ONLY ? lane & 15u : lane & 7u), y0 = oy + (int32_t)(DEPTH_ONLY ? lane >> 4 : lane >> 3);
    const double xd = (double)x0, yd = (double)y0;
#if ARCTIC_RASTER_I32
    if (DEPTH_ONLY && !RAW && (t.flags & RASTER_I32)) {   // wave-uniform (the record sits in scalar registers)
        // the three planes at the lane's first pixel in binary64 (exact integers below 2^31: plan_triangle), everything after that in 32-bit integers:
        // the pixels (x0, y0 + 4k) are 4 B[i] apart; (float) of the same integer, the same depth expression: the same bits as below
        int32_t v[3], sb[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) { v[i] = (int32_t)__builtin_fma(t.A[i], xd, __builtin_fma(t.B[i], yd, t.C[i])); sb[i] = (int32_t)t.B[i]; }
        const int32_t t0i = high_word(t.t0) != 0, t2i = high_word(t.t2) != 0;   // the thresholds are 0 or 1
        int32_t out_x = 0;
        if (cut) out_x = (x0 - fr.sc_x0) | (fr.sc_x1 - 1 - x0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k) { v[0] += sb[0] << 2; v[1] += sb[1] << 2; v[2] += sb[2] << 2; }
            int32_t outside = (v[0] - t0i) | v[1] | (v[2] - t2i);
            if (cut) { const int32_t py = y0 + 4 * k; outside |= out_x | (py - fr.sc_y0) | (fr.sc_y1 - 1 - py); }
            const float l1 = (float)v[2] * t.inv_area, l2 = (float)v[0] * t.inv_area;
            float z = fmaf(l2, t.dz2, fmaf(l1, t.dz1, t.z0));
            z = fminf(fmaxf(z, 0.0f), 1.0f);
            zb[k] = (outside >= 0 && z < 1.0f) ? __float_as_uint(z) : NO_DEPTH;
        }
        return;
    }
#endif
    double e[4][3];
    if (DEPTH_ONLY) {   // pixels (x0, y0 + 4k)
        const double yk[4] = {yd, yd + 4.0, yd + 8.0, yd + 12.0};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double v = __builtin_fma(t.A[i], xd, t.C[i]);
#pragma unroll
            for (int k = 0; k < 4; ++k) e[k][i] = __builtin_fma(t.B[i], yk[k], v);
        }
    } else {            // pixels (x0 + 8 (k & 1), y0 + 8 (k >> 1))
        const double xd8 = xd + 8.0, yd8 = yd + 8.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double u0 = __builtin_fma(t.B[i], yd, t.C[i]), u1 = __builtin_fma(t.B[i], yd8, t.C[i]);
            e[0][i] = __builtin_fma(t.A[i], xd, u0); e[1][i] = __builtin_fma(t.A[i], xd8, u0);
            e[2][i] = __builtin_fma(t.A[i], xd, u1); e[3][i] = __builtin_fma(t.A[i], xd8, u1);
        }
    }
    // the scissor test as sign bits, zero for blocks inside the scissor
    int32_t out_x[2] = {0, 0}, out_y[4] = {0, 0, 0, 0};
    if (cut) {
#pragma unroll
        for (int j = 0; j < (DEPTH_ONLY ? 1 : 2); ++j) { const int32_t px = x0 + 8 * j; out_x[j] = (px - fr.sc_x0) | (fr.sc_x1 - 1 - px); }
#pragma unroll
        for (int j = 0; j < (DEPTH_ONLY ? 4 : 2); ++j) { const int32_t py = y0 + (DEPTH_ONLY ? 4 : 8) * j; out_y[j] = (py - fr.sc_y0) | (fr.sc_y1 - 1 - py); }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float l1 = (float)e[k][2] * t.inv_area, l2 = (float)e[k][0] * t.inv_area;
        float z = fmaf(l2, t.dz2, fmaf(l1, t.dz1, t.z0));
        z = fminf(fmaxf(z, 0.0f), 1.0f);
        if (RAW) {
            const int32_t t0i = high_word(t.t0) != 0, t2i = high_word(t.t2) != 0;   // wave-uniform: 0 or 1
            const int32_t outside = (high_word(e[k][0]) - t0i) | high_word(e[k][1]) | (high_word(e[k][2]) - t2i)
                                  | out_x[DEPTH_ONLY ? 0 : k & 1] | out_y[DEPTH_ONLY ? k : k >> 1];
            zb[k] = __float_as_uint(z) | (uint32_t)(outside >> 31);   // NO_DEPTH where not covered
        } else {
            const int32_t outside = high_word(e[k][0] - t.t0) | high_word(e[k][1]) | high_word(e[k][2] - t.t2)
                                  | out_x[DEPTH_ONLY ? 0 : k & 1] | out_y[DEPTH_ONLY ? k : k >> 1];
            zb[k] = (outside >= 0 && z < 1.0f) ? __float_as_uint(z) : NO_DEPTH;   // depth LESS against the 1.0 clear
        }
    }
}

// where the four pixels of a lane live for block `code`: element index in the visibility plane / the depth map, and whether the
// (wave-uniform) tile row is this shard's
template <bool DEPTH_ONLY>
__device__ __forceinline__ void block_targets(uint32_t code, uint32_t lane, const RasterFrame &fr, uint32_t at[4], bool mine[4]) {
    const int32_t ox = (int32_t)(code & 0xFFFu) << 4, oy = (int32_t)((code >> 12) & 0xFFFu) << 4;
    if (DEPTH_ONLY) {
        const uint32_t x0 = (uint32_t)ox + (lane & 15u), y0 = (uint32_t)oy + (lane >> 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) { at[k] = (y0 + 4u * k) * (uint32_t)fr.pitch + x0; mine[k] = true; }
    } else {
        bool row_ok[2] = {true, true};
        uint32_t row_at[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ty_rel = ((oy >> 3) + j) - fr.tile_y0;
            int lrow = ty_rel;
            if (fr.band_tiles != 0) {   // interleaved shard: rows of other shards are skipped, the own ones are packed
                row_ok[j] = row_owned(ty_rel, fr.band_tiles, fr.shard_count, fr.shard_index);
                lrow = row_local(ty_rel, fr.band_tiles, fr.shard_count);
            }
            row_at[j] = ((uint32_t)lrow * (uint32_t)fr.tiles_x + (uint32_t)(ox >> 3)) * 64u;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) { at[k] = row_at[k >> 1] + (uint32_t)(k & 1) * 64u + lane; mine[k] = row_ok[k >> 1]; }
    }
}

// The same work item in 64-bit integers, for records with coordinates of 2^24 and more (triangles reaching far into the guard
// band of a 4K / 8K target): a 2x2 quad per lane; what depends only on record and block is wave-uniform and runs on the scalar
// unit, a lane adds its offset with 32x32+64-bit multiply-adds.  The same exact integers as edge_eval at every pixel.
template <bool DEPTH_ONLY>
__device__ __forceinline__ void raster_item_i64(const SetupRec &t, uint32_t code, uint32_t lane, const RasterFrame &fr,
                                                unsigned long long *__restrict__ vis, uint32_t *__restrict__ depth_bits) {
    const int32_t ox = (int32_t)(code & 0xFFFu) << 4, oy = (int32_t)((code >> 12) & 0xFFFu) << 4;
    const int32_t lx = (int32_t)(lane & 7) * 2, ly = (int32_t)(lane >> 3) * 2;                          // this lane's 2x2 quad in the block
    const int32_t qx = ox + lx, qy = oy + ly;
    if (qx > t.px1 || qx + 1 < t.px0 || qy > t.py1 || qy + 1 < t.py0) return;
    Edges e;
    make_edges(t, e);
    const float inv_area = 1.0f / (float)t.area2;
    int64_t eq[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
        eq[i] = edge_eval(e, i, ox, oy) + (int64_t)(int32_t)e.dx[i] * (int64_t)(ly * 256) - (int64_t)(int32_t)e.dy[i] * (int64_t)(lx * 256);
    // target row: both rows of the quad lie in the same 8-pixel tile row (qy is even)
    uint32_t row_base = 0;
    bool owned = true;
    if (!DEPTH_ONLY) {
        const int ty_rel = (qy >> 3) - fr.tile_y0;
        int lrow = ty_rel;
        if (fr.band_tiles != 0) {
            owned = row_owned(ty_rel, fr.band_tiles, fr.shard_count, fr.shard_index);
            lrow = row_local(ty_rel, fr.band_tiles, fr.shard_count);
        }
        row_base = (uint32_t)lrow * (uint32_t)fr.tiles_x;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int dx = k & 1, dy = k >> 1;
        const int32_t px = qx + dx, py = qy + dy;
        bool ok = owned && !(px < t.px0 || px > t.px1 || py < t.py0 || py > t.py1);   // the record's bounds are scissored
        const int64_t f0 = eq[0] + (dy ? e.dx[0] * 256 : 0) - (dx ? e.dy[0] * 256 : 0);
        const int64_t f1 = eq[1] + (dy ? e.dx[1] * 256 : 0) - (dx ? e.dy[1] * 256 : 0);
        const int64_t f2 = eq[2] + (dy ? e.dx[2] * 256 : 0) - (dx ? e.dy[2] * 256 : 0);
        ok = ok && !((f0 + e.bias[0]) < 0 || (f1 + e.bias[1]) < 0 || (f2 + e.bias[2]) < 0);
        const float l1 = (float)f2 * inv_area, l2 = (float)f0 * inv_area;
        float z = fmaf(l2, t.z[2] - t.z[0], fmaf(l1, t.z[1] - t.z[0], t.z[0]));
        z = fminf(fmaxf(z, 0.0f), 1.0f);
        ok = ok && (z < 1.0f);
        if (!ok) continue;
        if (DEPTH_ONLY) {
            uint32_t *p = depth_bits + ((uint32_t)py * (uint32_t)fr.pitch + (uint32_t)px);
            if (__float_as_uint(z) < *p) atomicMin(p, __float_as_uint(z));
        } else {
            unsigned long long *p = vis + ((row_base + ((uint32_t)px >> 3)) * 64u + ((uint32_t)py & 7u) * 8u + ((uint32_t)px & 7u));
            const unsigned long long key = ((unsigned long long)__float_as_uint(z) << 32) | t.order_id;
            if (key < *p) atomicMin(p, key);
        }
    }
}

// ascending bitonic sort of one (key, value) pair per lane across the wave
__device__ __forceinline__ void wave_sort(uint32_t &key, uint32_t &val, uint32_t lane) {
#pragma unroll
    for (uint32_t k = 2; k <= 64; k <<= 1)
#pragma unroll
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            const uint32_t pk = __shfl_xor(key, (int)j), pv = __shfl_xor(val, (int)j);
            const bool keep_low = ((lane & j) == 0) == ((lane & k) == 0);   // this lane keeps the smaller of the pair
            const bool take = keep_low ? pk < key : pk > key;
            if (take) { key = pk; val = pv; }
        }
}

// A wave's accumulator for one 16x16 block: the smallest key seen per pixel, four pixels per lane.  Forward pass: key = depth
// bits << 32 | order id (ties: first drawn wins); shadow pass: the depth bits.
template <bool DEPTH_ONLY> struct KeyOf { typedef unsigned long long type; };
template <> struct KeyOf<true> { typedef uint32_t type; };

// Persistent: the number of work items is only known on the device (counters[1]), so a fixed grid strides over the item table
// and the host never waits for a count.
// What bounds a rasteriser that sends every covered pixel to memory on its own (tools/experiments/atomic_rates.hip, profiles/):
// the memory side retires ~25 atomic requests per ns -- a request = the lanes of one instruction that fall into one 64-byte
// segment, whatever their number, workgroup or agent scope, 32 or 64 bit -- and a 4K frame issued 1.65 M of them (5.8 of 8
// pixels per request: triangle edges and overdraw): 66 us of 85.  So requests are merged before they leave the wave: a wave takes
// CHUNK consecutive work items (neighbouring triangles of a mesh), sorts them by block (bitonic sort across the lanes), and walks
// the runs of equal block with the block's 256 keys in registers (four per lane): an item costs its plane evaluations and a
// min per pixel, a run costs ONE early depth read and ONE atomicMin per pixel that still wins.  The memory side of a run is
// split around the next run: reads issued when the run ends, atomics when the next one ends -- the reads' latency, and the wait
// for the previous atomics that comes with it (loads, stores and atomics retire through one in-order counter), hide behind the
// next run's arithmetic.
#ifndef RASTER_CHUNK
#define RASTER_CHUNK 32
#endif
template <bool DEPTH_ONLY, uint32_t CHUNK>
__global__ __launch_bounds__(256) void k_raster(const SetupRec *__restrict__ recs, const RasterRec *__restrict__ rrecs,
                                                const uint2 *__restrict__ items, uint32_t item_cap,
                                                const uint32_t *__restrict__ counters, const GeomParams gp,
                                                unsigned long long *__restrict__ vis, uint32_t *__restrict__ depth_bits,
                                                uint32_t *__restrict__ host_counts, uint32_t *__restrict__ host_overflow, uint32_t after_owned) {
    prepass_priority();
    typedef typename KeyOf<DEPTH_ONLY>::type Key;
    constexpr Key NONE = (Key)~(Key)0;
    // records, work items and the overflow flag for the host (pinned, mapped memory: no copy launches); read after a synchronise
    if (blockIdx.x == 0 && threadIdx.x == 0) { host_counts[0] = counters[0]; host_counts[1] = counters[1]; *host_overflow = counters[2]; host_overflow[2] = after_owned ? counters[4] : counters[1]; }
    if (counters[2]) return;   // a table overflowed in k_setup: entries are missing, the host reports the frame as dropped
    // after the block owners: `items` is what the bins did not take (k_bin), usually nothing
    const uint32_t n_items = min(after_owned ? counters[4] : counters[1], item_cap);
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), n_waves = gridDim.x * 4;
    const RasterFrame fr = {gp.sc_x0, gp.sc_y0, gp.sc_x1, gp.sc_y1, gp.tiles_x, gp.tile_y0, gp.pitch, gp.band_tiles, gp.shard_index, gp.shard_count};
    // the finished run whose reads are in flight: its keys, where they go, what the target holds
    bool pending = false;
    Key pend_key[4] = {NONE, NONE, NONE, NONE}, pend_cur[4] = {0, 0, 0, 0};
    uint32_t pend_at[4] = {0, 0, 0, 0};
    const auto write_pending = [&]() {
        if (!pending) return;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (pend_key[k] != NONE && pend_key[k] < pend_cur[k]) {
                if (DEPTH_ONLY) __hip_atomic_fetch_min((uint32_t __attribute__((address_space(1))) *)((gbytes)depth_bits + pend_at[k] * 4u), (uint32_t)pend_key[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else __hip_atomic_fetch_min((unsigned long long __attribute__((address_space(1))) *)((gbytes)vis + pend_at[k] * 8u), (unsigned long long)pend_key[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        pending = false;
    };
    // byte offsets fit 32 bits (the planes are at most 16384^2 entries of 8 / 4 bytes): wave-uniform base + per-lane offset
    const auto read_run = [&](const Key acc[4], uint32_t code) {
        bool mine[4];
        block_targets<DEPTH_ONLY>(code, lane, fr, pend_at, mine);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            pend_key[k] = mine[k] ? acc[k] : NONE;
            pend_cur[k] = 0;
            if (pend_key[k] != NONE) {
                if (DEPTH_ONLY) pend_cur[k] = (Key) * (const uint32_t __attribute__((address_space(1))) *)((gbytes)depth_bits + pend_at[k] * 4u);
                else pend_cur[k] = (Key) * (const unsigned long long __attribute__((address_space(1))) *)((gbytes)vis + pend_at[k] * 8u);
            }
        }
        pending = true;
    };
    // few items: shorter chunks, so that every resident wave has one (a 512^2 frame is 20 k items for 6 k waves)
    const uint32_t chunk = min(CHUNK, max(1u, (n_items + n_waves - 1) / n_waves));
    for (uint32_t first = wave * chunk; first < n_items; first += n_waves * chunk) {   // uniform per wave
        uint32_t code = 0xFFFFFFFFu, rec = ITEM_SKIP;
        if (lane < chunk && first + lane < n_items) {
            const uint2 en = items[first + lane];
            rec = en.x;
            if (rec != ITEM_SKIP) code = en.y;
        }
        const uint32_t n_live = (uint32_t)__popcll(__ballot(rec != ITEM_SKIP));
        if (n_live == 0) continue;    // (after k_bin: most chunks)
        wave_sort(code, rec, lane);   // skipped entries sort to the end
        Key acc[4] = {NONE, NONE, NONE, NONE};
        uint32_t run_code = __builtin_amdgcn_readlane(code, 0);
        // Shadow pass (instruction and latency bound: 72.5 -> 69.0 us at 4000^2): the record of item i + 1 is requested when that of
        // item i has arrived, before item i is evaluated (as in k_raster_owned: the empty asm pins the one wait scalar loads allow
        // in front of the next request).  The forward pass through this kernel is atomic bound and loses 2 us to it: there the record
        // is fetched when it is needed.
        uint32_t r_next = __builtin_amdgcn_readlane(rec, 0);
        RasterRec q_next;
        if (DEPTH_ONLY) q_next = rrecs[r_next];
        for (uint32_t i = 0; i < n_live; ++i) {
            const uint32_t c = __builtin_amdgcn_readlane(code, i), r = DEPTH_ONLY ? r_next : __builtin_amdgcn_readlane(rec, i);
            if (DEPTH_ONLY) asm volatile("" :: "s"(q_next.flags) : "memory");
            const RasterRec q = DEPTH_ONLY ? q_next : rrecs[r];
            if (DEPTH_ONLY && i + 1 < n_live) { r_next = __builtin_amdgcn_readlane(rec, i + 1); q_next = rrecs[r_next]; }
            if (c != run_code) {   // the run is complete: write the one before it, start this one's reads
                write_pending();
                read_run(acc, run_code);
                acc[0] = acc[1] = acc[2] = acc[3] = NONE;
                run_code = c;
            }
            if (q.flags & RASTER_EXACT_F64) {
                uint32_t zb[4];
                item_pixels<DEPTH_ONLY>(q, c, lane, fr, zb);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const Key key = DEPTH_ONLY ? (Key)zb[k] : (zb[k] == NO_DEPTH ? NONE : (Key)(((unsigned long long)zb[k] << 32) | q.order_id));
                    acc[k] = key < acc[k] ? key : acc[k];
                }
            } else raster_item_i64<DEPTH_ONLY>(recs[r], c, lane, fr, vis, depth_bits);   // rare: the record's own integer path, straight to memory
        }
        write_pending();
        read_run(acc, run_code);
    }
    write_pending();
}

// ---------------------------------------------------------------------------------------------
// Block ownership: no per-pixel atomics (round 3).  The atomic rasteriser above is bound by the memory side retiring one request
// per 64-byte segment and instruction (25 per ns: 59 of its 72 us at 4K), and merging inside a wave only reaches the items a wave
// happens to hold.  Here every 16x16 block of the target has ONE owner: k_bin hands each work item to the bin of its block (one
// returning atomic per ITEM on the block's counter -- 160 k per 4K frame instead of 1.47 M per-pixel requests), then one wave per
// block evaluates its bin with the block's 256 keys in registers and stores them once: that store is also the clear.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * ARCTIC_RASTER_WGW) void k_bin(const uint2 *__restrict__ items, uint2 *__restrict__ left, uint32_t item_cap, uint32_t *__restrict__ counters, const BinTables B) {
    prepass_priority();
    if (counters[2]) return;
    const uint32_t n_items = min(counters[1], item_cap);
    const uint32_t lane = threadIdx.x & 63;
    for (uint32_t first = blockIdx.x * (64u * ARCTIC_RASTER_WGW) + (threadIdx.x & ~63u); first < n_items; first += gridDim.x * (64u * ARCTIC_RASTER_WGW)) {   // uniform per wave
        const uint32_t i = first + lane;
        bool keep = false;     // not binned: the atomic rasteriser draws it
        uint2 en = make_uint2(ITEM_SKIP, 0u);
        if (i < n_items) en = items[i];
        if (en.x != ITEM_SKIP) {
            keep = (en.y & ITEM_INEXACT) != 0;
            if (!keep) {
                const uint32_t b = ((en.y >> 12) & 0xFFFu) * B.blocks_x + (en.y & 0xFFFu);
                // a full bin is not asked again (the count only grows, so a stale read errs on the side of asking): a dense mesh in
                // a few blocks would otherwise queue thousands of atomics on one address
                keep = !B.count_all && __builtin_nontemporal_load(&B.count[b]) >= BIN_SLOTS;
                if (!keep) {
                    const uint32_t slot = atomicAdd(&B.count[b], 1u);
                    if (slot < BIN_SLOTS) B.slots[b * BIN_SLOTS + slot] = en.x; else keep = true;
                }
            }
        }
        const unsigned long long m = __ballot(keep);
        if (m) {   // rare: compacted behind one atomic per wave
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&counters[4], (uint32_t)__popcll(m));
            base = __shfl(base, 0);
            if (keep) left[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = en;
        }
    }
}

// MERGE (the shadow pass beside k_setup's small-triangle path, round 5): the target was cleared and the small triangles are in it already -- an owner starts from
// what its block holds instead of from nothing, and a block whose bin is empty (two thirds of a sun's map) has no owner at all
template <bool DEPTH_ONLY, bool MERGE = false>
__global__ __launch_bounds__(64 * ARCTIC_RASTER_WGW) void k_raster_owned(const RasterRec *__restrict__ rrecs, const BinTables B, const GeomParams gp,
                                                      unsigned long long *__restrict__ vis, uint32_t *__restrict__ depth_bits) {
    prepass_priority();
    typedef typename KeyOf<DEPTH_ONLY>::type Key;
    constexpr Key NONE = (Key)~(Key)0;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t w = blockIdx.x * ARCTIC_RASTER_WGW + (ARCTIC_RASTER_WGW == 1 ? 0u : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6));
    if (w >= B.grid_x * B.grid_y) return;
    const uint32_t bx = w % B.grid_x, gy = w / B.grid_x;
    const uint32_t by = owner_block_row(gy, B.by0, B.local_rows, gp.band_tiles, gp.shard_count, gp.shard_index, gp.tile_y0);   // block row of the target
    const RasterFrame fr = {gp.sc_x0, gp.sc_y0, gp.sc_x1, gp.sc_y1, gp.tiles_x, gp.tile_y0, gp.pitch, gp.band_tiles, gp.shard_index, gp.shard_count};
    const int32_t ox = (int32_t)bx * 16, oy = (int32_t)by * 16;
    const bool whole = ox >= gp.sc_x0 && ox + 16 <= gp.sc_x1 && oy >= gp.sc_y0 && oy + 16 <= gp.sc_y1;
    const uint32_t code = bx | (by << 12) | (whole ? 0u : ITEM_SCISSOR);
    // where the block's pixels live, and which of them this shard stores
    uint32_t at[4]; bool mine[4];
    block_targets<DEPTH_ONLY>(code, lane, fr, at, mine);
    if (DEPTH_ONLY) {
        const int32_t x0 = ox + (int32_t)(lane & 15u), y0 = oy + (int32_t)(lane >> 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) mine[k] = x0 < gp.sc_x1 && y0 + 4 * k >= gp.sc_y0 && y0 + 4 * k < gp.sc_y1;   // (sc_x0 = 0: the map's rows, or this rank's slice of them)
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int lrow;
            mine[k] = owner_tile_row(by, k >> 1, gp.tile_y0, gp.tiles_y, gp.band_tiles, gp.shard_count, gp.shard_index, lrow) && (ox >> 3) + (k & 1) < gp.tiles_x;
        }
    }
    const uint32_t b = by * B.blocks_x + bx;
    // the bin: its counter through the scalar unit, its slots one per lane -- all BIN_SLOTS of them, used or not, so that the
    // load does not wait for the counter
    const uint32_t my = lane < BIN_SLOTS ? B.slots[b * BIN_SLOTS + lane] : 0u;
    const uint32_t n = min(B.count[b], BIN_SLOTS);   // wave-uniform (scalar load)
    if (MERGE && n == 0) return;
    // keys with depth bits of 0x3F800000 (a covered pixel at depth 1.0: fails LESS against the clear) or 0xFFFFFFFF (not covered)
    // lose against every drawn pixel in the merge and are turned into "nothing drawn" once, at the end
    Key acc[4] = {NONE, NONE, NONE, NONE};
    if (MERGE) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (mine[k]) acc[k] = DEPTH_ONLY ? (Key) * (const uint32_t __attribute__((address_space(1))) *)((gbytes)depth_bits + at[k] * 4u)
                                             : (Key) * (const unsigned long long __attribute__((address_space(1))) *)((gbytes)vis + at[k] * 8u);
    }
    const auto merge = [&](const RasterRec &q) {
        uint32_t zb[4];
        item_pixels<DEPTH_ONLY, true>(q, code, lane, fr, zb);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const Key key = DEPTH_ONLY ? (Key)zb[k] : (Key)(((unsigned long long)zb[k] << 32) | q.order_id);
            acc[k] = key < acc[k] ? key : acc[k];
        }
    };
    // the record of item i + 1 is requested when that of item i has arrived, i.e. before item i is evaluated: scalar loads return
    // out of order, so the only wait there is waits for all of them -- the empty asm (a use of the record, a compiler barrier for
    // memory operations) pins that wait in front of the next request instead of behind it
    if (n) {
        RasterRec q0 = rrecs[__builtin_amdgcn_readlane(my, 0)], q1 = q0;
        for (uint32_t i = 0;;) {
            asm volatile("" :: "s"(q0.flags) : "memory");
            if (i + 1 < n) q1 = rrecs[__builtin_amdgcn_readlane(my, i + 1)];
            merge(q0);
            if (++i == n) break;
            asm volatile("" :: "s"(q1.flags) : "memory");
            if (i + 1 < n) q0 = rrecs[__builtin_amdgcn_readlane(my, i + 1)];
            merge(q1);
            if (++i == n) break;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (!mine[k]) continue;
        if (DEPTH_ONLY) *(uint32_t __attribute__((address_space(1))) *)((gbytes)depth_bits + at[k] * 4u) = min((uint32_t)acc[k], 0x3F800000u);   // depth 1.0 where nothing was drawn
        else *(unsigned long long __attribute__((address_space(1))) *)((gbytes)vis + at[k] * 8u) = (uint32_t)(acc[k] >> 32) >= 0x3F800000u ? ~0ull : (unsigned long long)acc[k];
    }
}

// ---------------------------------------------------------------------------------------------
// resolve: visibility -> interpolated attributes, tile-major G-buffer
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_resolve(const unsigned long long *__restrict__ vis, const SetupRec *__restrict__ recs, const RasterRec *__restrict__ rrecs,
                                                 const uint32_t *__restrict__ rec_of, const ObjectRec *__restrict__ objs, const XVert *__restrict__ xv,
                                                 const GeomParams gp, uint32_t n_tiles, GBuffer g, const TileHint hint) {
    uint32_t tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= n_tiles) return;
    uint32_t lane = threadIdx.x & 63;
    size_t idx = (size_t)tile * 64 + lane;
    unsigned long long key = vis[idx];
    float a[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) a[k] = 0.0f;
    uint32_t mat = NO_MATERIAL;
    if (key != ~0ull) {
        const uint32_t ri = rec_of[(uint32_t)key];   // low word = order id (k_setup)
        const SetupRec &t = recs[ri];
        int32_t tx = (int32_t)(tile % (uint32_t)gp.tiles_x);
        int32_t ty = row_global((int)(tile / (uint32_t)gp.tiles_x), gp.band_tiles, gp.shard_count, gp.shard_index) + gp.tile_y0;
        int32_t px = tx * 8 + (int32_t)(lane & 7), py = ty * 8 + (int32_t)(lane >> 3);
        float B[3];
        source_barycentrics(t, rrecs[ri], px, py, B);
        const ObjectRec &ob = objs[t.object];
        uint32_t lt = t.src_tri - ob.first_triangle;
        const float *A0 = xv[ob.first_xvert + ob.indices[3 * lt]].attr;
        const float *A1 = xv[ob.first_xvert + ob.indices[3 * lt + 1]].attr;
        const float *A2 = xv[ob.first_xvert + ob.indices[3 * lt + 2]].attr;
#pragma unroll
        for (int k = 0; k < 18; ++k) a[k] = interpolate_attr(B, A0, A1, A2, k);
        mat = ob.material;
    }
    {
        float4 A, C, D, E; float B3[3];
        gbuffer_pack(a, mat, A, B3, C, D, E);
        g.a[idx] = A; g.c[idx] = C; g.d[idx] = D; g.e[idx] = E;
        g.b[idx * 3] = B3[0]; g.b[idx * 3 + 1] = B3[1]; g.b[idx * 3 + 2] = B3[2];
    }
    // The tile's COST CLASS, a hint for the shading pass's dispatch order (k_tile_order; results never depend on it): 1 when a pixel
    // of the tile can be lit at all -- covered, and not decided "every tap shadowed" by the shadow map's min/max table, the first
    // test of the shading kernel itself (shade.hip: shadow_quick) -- or takes the environment lookup; such a tile runs the light loop,
    // tens of times the work of a tile in full shadow.  The light-space position is in registers here: the hint costs one 8-byte load.
    if (hint.tile_class) {
        bool costly = mat != NO_MATERIAL || hint.sky != 0;
        if (mat != NO_MATERIAL && hint.bounds) {
            ShadowPos p;
            shadow_coords(a[14], a[15], a[16], a[17], p);
            uint32_t offset;
            if (shadow_table_offset(hint.S, hint.pitch, p, offset)) {
                const float2 mm = *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(hint.bounds) + offset);
                costly = !(p.pz > mm.y);
            }
        }
        const unsigned long long any = __ballot(costly);
        if (lane == 0) hint.tile_class[tile] = any ? 1 : 0;
    }
}

// ---------------------------------------------------------------------------------------------
// the shading pass's dispatch order from the tiles' cost classes: EIGHT workgroups, one per XCD's share of the tile rows
// ---------------------------------------------------------------------------------------------
// A job of the shading pass = a STRIP of 4 horizontally adjacent tiles (one workgroup, a wave per tile), coded ty << 16 | strip
// column.  A strip is costly when one of its tiles is.  Round 4 built ONE list over the whole frame in one workgroup (112 us at 4K, and handing
// strips out by cost alone sent neighbouring strips to different XCDs: +83 MB of fabric reads per pass, profiles/r5_a_traffic_tile_order.json).
// Round 5: the strips of tile rows ty = x (mod 8) form list x -- the rows the geometric order gives to "XCD x" -- and each list is ordered
// by its own workgroup, independently (no scan across workgroups): costly strips dealt evenly over the first (1 - tail) of the list, in raster
// order, everything else filled with the cheap ones in raster order; position p of a list holds a costly strip iff cnt(p + 1) > cnt(p),
// cnt(p) = ceil(p nL / span) = costly strips in front of p.  The lists are interleaved in groups of G = tiles per wave (common.h order_slot):
// block b of the pass takes the slots b G ... b G + G - 1, all of list b % 8 -- blocks are dealt round-robin over the XCDs, so a list stays
// on one XCD and horizontally adjacent strips still meet in one L2.  Lists shorter than the longest end in ORDER_NONE slots (skipped by the pass).
__device__ __forceinline__ uint32_t costly_before(uint32_t p, uint32_t nL, uint32_t span) {
    const unsigned long long c = ((unsigned long long)p * nL + span - 1) / span;
    return c < nL ? (uint32_t)c : nL;
}
__global__ __launch_bounds__(1024) void k_tile_order(const uint8_t *__restrict__ tile_class, uint32_t tiles_x, uint32_t tiles_y, uint32_t tail_permille, uint32_t group,
                                                     uint32_t *__restrict__ lists /* 2 N: per list its costly strips, then its cheap strips */, uint32_t *__restrict__ order /* order_slots() */) {
    __shared__ uint32_t wave_count[16];
    __shared__ uint32_t base_costly;
    const uint32_t x = blockIdx.x, t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t bpr = (tiles_x + 3) / 4;
    const uint32_t rows = tiles_y > x ? (tiles_y - x + 7) / 8 : 0u, N = bpr * rows;        // this list: tile rows x, x + 8, ...
    const uint32_t rows_max = (tiles_y + 7) / 8, L = (bpr * rows_max + group - 1) / group * group;   // slots per list (the longest list, whole groups)
    // where this list's scratch starts: lists 0 .. x - 1 hold bpr * their rows entries each, twice (costly | cheap)
    uint32_t before = 0;
    for (uint32_t y = 0; y < x; ++y) before += tiles_y > y ? (tiles_y - y + 7) / 8 : 0u;
    uint32_t *costly_list = lists + 2u * bpr * before, *cheap_list = costly_list + N;
    const auto code_of = [&](uint32_t s) { return ((8u * (s / bpr) + x) << 16) | (s % bpr); };
    const auto costly = [&](uint32_t s) {
        const uint32_t ty = 8u * (s / bpr) + x, x0 = (s % bpr) * 4, x1 = min(tiles_x, x0 + 4);
        uint32_t c = 0;
        for (uint32_t xx = x0; xx < x1; ++xx) c |= tile_class[(size_t)ty * tiles_x + xx];
        return c != 0;
    };
    if (t == 0) base_costly = 0;
    __syncthreads();
    // compaction in raster order, 1024 strips at a time: rank in the wave by ballot, the waves' counts through LDS
    for (uint32_t cb = 0; cb < N; cb += 1024) {
        const uint32_t s = cb + t;
        const bool in = s < N, c = in && costly(s);
        const unsigned long long m = __ballot(c);
        if (lane == 0) wave_count[wave] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t pre = 0, tot = 0;
        for (uint32_t w = 0; w < 16; ++w) { const uint32_t n = wave_count[w]; pre += w < wave ? n : 0u; tot += n; }
        const uint32_t rank_c = base_costly + pre + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (in) {
            if (c) costly_list[rank_c] = code_of(s);
            else cheap_list[s - rank_c] = code_of(s);     // cheap strips in front of s = s - costly strips in front of s
        }
        __syncthreads();
        if (t == 0) base_costly += tot;
        __syncthreads();
    }
    __threadfence_block();
    __syncthreads();
    const uint32_t nL = base_costly;
    const uint32_t tail = (uint32_t)((unsigned long long)N * tail_permille / 1000);
    const uint32_t span = max(max(nL, N - min(N, tail)), 1u);
    for (uint32_t p = t; p < L; p += 1024) {
        uint32_t e = ORDER_NONE;
        if (p < N) {
            const uint32_t c = costly_before(p, nL, span);
            e = costly_before(p + 1, nL, span) > c ? costly_list[c] : cheap_list[p - c];
        }
        order[order_slot(x, p, group)] = e;
    }
}

// ---------------------------------------------------------------------------------------------
// helpers: fills; row-major <-> tile-major G-buffer conversion for the read/write test entry points
// ---------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void k_fill(T *p, T v, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) p[i] = v;
}

__global__ __launch_bounds__(256) void k_gbuffer_tile(GBuffer g, float *attrs, uint32_t *mat, uint32_t width, uint32_t rows,
                                                      uint32_t row0_in_tile, uint32_t tiles_x, uint32_t tiles_y, int to_tiled) {
    uint32_t tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= tiles_x * tiles_y) return;
    uint32_t lane = threadIdx.x & 63;
    uint32_t x = (tile % tiles_x) * 8 + (lane & 7);
    int32_t y = (int32_t)((tile / tiles_x) * 8 + (lane >> 3)) - (int32_t)row0_in_tile;   // row inside the shard
    size_t idx = (size_t)tile * 64 + lane;
    bool in = x < width && y >= 0 && y < (int32_t)rows;
    if (to_tiled) {
        float a[18];
        uint32_t m = NO_MATERIAL;
#pragma unroll
        for (int k = 0; k < 18; ++k) a[k] = 0.0f;
        if (in) {
            size_t p = (size_t)y * width + x;
#pragma unroll
            for (int k = 0; k < 18; ++k) a[k] = attrs[p * 18 + k];
            m = mat[p];
        }
        float4 A, C, D, E; float B3[3];
        gbuffer_pack(a, m, A, B3, C, D, E);
        g.a[idx] = A; g.c[idx] = C; g.d[idx] = D; g.e[idx] = E;
        g.b[idx * 3] = B3[0]; g.b[idx * 3 + 1] = B3[1]; g.b[idx * 3 + 2] = B3[2];
    } else if (in) {
        size_t p = (size_t)y * width + x;
        float4 A = g.a[idx], C = g.c[idx], D = g.d[idx], E = g.e[idx];
        float b0 = g.b[idx * 3], b1 = g.b[idx * 3 + 1], b2 = g.b[idx * 3 + 2];
        if (attrs) {
            float *a = attrs + p * 18;
            a[0] = A.x; a[1] = A.y; a[2] = C.w; a[3] = D.x; a[4] = D.y; a[5] = D.z; a[6] = D.w;
            a[7] = E.x; a[8] = E.y; a[9] = E.z; a[10] = E.w;
            a[11] = C.x; a[12] = C.y; a[13] = C.z;
            a[14] = A.z; a[15] = A.w; a[16] = b0; a[17] = b1;
        }
        if (mat) mat[p] = __float_as_uint(b2);
    }
}

inline uint32_t div_up(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

}  // namespace

hipError_t launch_vertex(const ObjectRec *objs, const uint32_t *block_obj, const uint32_t *block_first, uint32_t n_blocks,
                         const GeomParams &gp, XVert *xv, int clip_only, uint32_t *counters, unsigned long long *clear, unsigned long long clear_value,
                         size_t clear_count, uint32_t *zero, size_t zero_count, const float *block_bounds, hipStream_t s) {
    if (n_blocks == 0) return clear_count ? launch_fill_u64(clear, clear_value, clear_count, s) : hipSuccess;
    k_vertex<<<n_blocks, 256, 0, s>>>(objs, block_obj, block_first, gp, xv, clip_only, counters, clear, clear_value, clear_count, zero, zero_count, block_bounds);
    return hipGetLastError();
}

hipError_t launch_setup(const ObjectRec *objs, const uint32_t *block_obj, const uint32_t *block_first, uint32_t n_blocks,
                        const GeomParams &gp, const XVert *xv, SetupRec *recs, RasterRec *rrecs, uint32_t *rec_of, uint2 *items, uint32_t item_cap,
                        uint32_t rec_cap, uint32_t *counters, uint2 *clip_list, const float *block_bounds, uint32_t *small_depth_bits, hipStream_t s) {
    if (n_blocks == 0) return hipSuccess;
    const SetupTables T = {recs, rrecs, rec_of, items, item_cap, rec_cap, counters, small_depth_bits};
    // small_depth_bits: the shadow pass's map when its small triangles are to be drawn by k_setup (never with the integer path forced: that switch is an A/B of the item rasterisers)
    if (small_depth_bits && SMALL_PX > 0 && !(gp.raster_flags & 1)) k_setup<true><<<n_blocks, SETUP_THREADS, 0, s>>>(objs, block_obj, block_first, gp, xv, T, clip_list, block_bounds);
    else k_setup<false><<<n_blocks, SETUP_THREADS, 0, s>>>(objs, block_obj, block_first, gp, xv, T, clip_list, block_bounds);
    // the clip list's length stays on the device: a fixed small grid strides over it (empty in most frames of most scenes)
    k_setup_clipped<<<std::min<uint32_t>(n_blocks * (SETUP_THREADS / CLIP_LANES), 256u), 64, 0, s>>>(objs, gp, xv, T, clip_list);
    return hipGetLastError();
}

// the persistent grid of k_raster: exactly the waves that are resident at once (a grid larger than that runs its last blocks
// after the others have finished their share of the items)
uint32_t raster_grid_blocks(bool depth_only, uint32_t cu_count) {
    int per_cu = 0;
    const hipError_t e = depth_only ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_raster<true, RASTER_CHUNK>, 256, 0)
                                    : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_raster<false, RASTER_CHUNK>, 256, 0);
    if (e != hipSuccess || per_cu < 1) { (void)hipGetLastError(); per_cu = 4; }
    return cu_count * (uint32_t)per_cu;
}

hipError_t launch_raster_vis(const SetupRec *recs, const RasterRec *rrecs, const uint2 *items, uint32_t item_cap, const uint32_t *counters, uint32_t grid_blocks,
                             const GeomParams &gp, unsigned long long *vis, uint32_t *host_counts, uint32_t *host_overflow, bool after_owned, hipStream_t s) {
    k_raster<false, RASTER_CHUNK><<<grid_blocks, 256, 0, s>>>(recs, rrecs, items, item_cap, counters, gp, vis, nullptr, host_counts, host_overflow, after_owned ? 1u : 0u);
    return hipGetLastError();
}

hipError_t launch_raster_owned(bool depth_only, const RasterRec *rrecs, const uint2 *items, uint2 *left, uint32_t item_cap, uint32_t *counters, const BinTables &B,
                               const GeomParams &gp, unsigned long long *vis, uint32_t *depth_bits, bool merge, hipStream_t s) {
    // the item count stays on the device: a fixed grid strides over the table
    k_bin<<<std::min<uint32_t>(div_up(item_cap, 64u * ARCTIC_RASTER_WGW), 4096u / ARCTIC_RASTER_WGW), 64 * ARCTIC_RASTER_WGW, 0, s>>>(items, left, item_cap, counters, B);
    const uint32_t n = B.grid_x * B.grid_y;
    if (n) {
        if (depth_only && merge) k_raster_owned<true, true><<<div_up(n, ARCTIC_RASTER_WGW), 64 * ARCTIC_RASTER_WGW, 0, s>>>(rrecs, B, gp, vis, depth_bits);
        else if (depth_only) k_raster_owned<true><<<div_up(n, ARCTIC_RASTER_WGW), 64 * ARCTIC_RASTER_WGW, 0, s>>>(rrecs, B, gp, vis, depth_bits);
        else k_raster_owned<false><<<div_up(n, ARCTIC_RASTER_WGW), 64 * ARCTIC_RASTER_WGW, 0, s>>>(rrecs, B, gp, vis, depth_bits);
    }
    return hipGetLastError();
}

hipError_t launch_raster_depth(const SetupRec *recs, const RasterRec *rrecs, const uint2 *items, uint32_t item_cap, const uint32_t *counters, uint32_t grid_blocks,
                               const GeomParams &gp, uint32_t *depth_bits, uint32_t *host_counts, uint32_t *host_overflow, bool after_owned, hipStream_t s) {
    k_raster<true, RASTER_CHUNK><<<grid_blocks, 256, 0, s>>>(recs, rrecs, items, item_cap, counters, gp, nullptr, depth_bits, host_counts, host_overflow, after_owned ? 1u : 0u);
    return hipGetLastError();
}

hipError_t launch_resolve(const unsigned long long *vis, const SetupRec *recs, const RasterRec *rrecs, const uint32_t *rec_of, const ObjectRec *objs, const XVert *xv,
                          const GeomParams &gp, uint32_t n_tiles, GBuffer g, const TileHint &hint, hipStream_t s) {
    if (n_tiles == 0) return hipSuccess;
    k_resolve<<<div_up(n_tiles, 4), 256, 0, s>>>(vis, recs, rrecs, rec_of, objs, xv, gp, n_tiles, g, hint);
    return hipGetLastError();
}
hipError_t launch_tile_order(const uint8_t *tile_class, uint32_t tiles_x, uint32_t tiles_y, uint32_t tail_permille, uint32_t group, uint32_t *lists, uint32_t *order, hipStream_t s) {
    if (tiles_x == 0 || tiles_y == 0 || group == 0) return hipSuccess;
    k_tile_order<<<8, 1024, 0, s>>>(tile_class, tiles_x, tiles_y, tail_permille, group, lists, order);
    return hipGetLastError();
}

hipError_t launch_fill_u64(unsigned long long *p, unsigned long long v, size_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    k_fill<unsigned long long><<<(unsigned)std::min<size_t>((n + 255) / 256, 2048), 256, 0, s>>>(p, v, n);
    return hipGetLastError();
}
hipError_t launch_fill_u32(uint32_t *p, uint32_t v, size_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    k_fill<uint32_t><<<(unsigned)std::min<size_t>((n + 255) / 256, 2048), 256, 0, s>>>(p, v, n);
    return hipGetLastError();
}

hipError_t launch_gbuffer_tile(GBuffer g, float *attrs, uint32_t *mat, uint32_t width, uint32_t rows, uint32_t row0_in_tile,
                               uint32_t tiles_x, uint32_t tiles_y, int to_tiled, hipStream_t s) {
    k_gbuffer_tile<<<div_up(tiles_x * tiles_y, 4), 256, 0, s>>>(g, attrs, mat, width, rows, row0_in_tile, tiles_x, tiles_y, to_tiled);
    return hipGetLastError();
}

}  // namespace arctic
