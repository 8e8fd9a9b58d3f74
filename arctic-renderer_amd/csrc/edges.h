// edges.h -- the integer edge functions shared by the rasteriser (geometry.hip) and the visibility-buffer shading kernel
// (shade.hip): device code only.
#pragma once
#include "common.h"

namespace arctic {

// edge i: vertex i -> vertex (i+1)%3, inside-positive; bias implements the top-left rule
struct Edges {
    int64_t dx[3], dy[3];
    int32_t x0[3], y0[3];
    int64_t bias[3];
};
__device__ __forceinline__ void make_edges(const SetupRec &t, Edges &e) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        int j = (i + 1) % 3;
        e.dx[i] = (int64_t)t.X[j] - t.X[i];
        e.dy[i] = (int64_t)t.Y[j] - t.Y[i];
        e.x0[i] = t.X[i]; e.y0[i] = t.Y[i];
        bool top_left = (e.dy[i] == 0 && e.dx[i] > 0) || (e.dy[i] < 0);
        e.bias[i] = top_left ? 0 : -1;
    }
}
// All four factors fit 32 bits: the coordinates are 24.8 fixed point inside the +-64 w guard band of a target of at most
// 16 k pixels (|X| < 2^29), pixel centres are below 2^23 -- so the exact 64-bit value costs two 32x32->64 multiply-adds
// (v_mad_i64_i32) instead of two full 64-bit multiplications.
__device__ __forceinline__ int64_t edge_eval(const Edges &e, int i, int32_t px, int32_t py) {
    const int32_t Px = px * 256 + 128, Py = py * 256 + 128;
    return (int64_t)(int32_t)e.dx[i] * (int64_t)(Py - e.y0[i]) - (int64_t)(int32_t)e.dy[i] * (int64_t)(Px - e.x0[i]);
}

// perspective-correct barycentrics of pixel (px, py) with respect to the SOURCE triangle of record t (the weights the 18
// VSOut attributes are interpolated with): screen-space barycentrics from the edge functions, 1/w correction, then through
// the record's own barycentric coordinates in its source triangle, so clipped triangles need no vertices of their own.
// Every operation rounds once, in this order, wherever it is compiled (the callers disable fp contraction).
// The two edge functions come from the record's binary64 planes (RasterRec: exact integers, the same values as edge_eval) unless
// the record's coordinates are too large for those (RASTER_EXACT_F64 clear: the 64-bit integer path).
__device__ __forceinline__ void source_barycentrics(const SetupRec &t, const RasterRec &q, int32_t px, int32_t py, float B[3]) {
#pragma clang fp contract(off)
    float l1, l2;
    if (q.flags & RASTER_EXACT_F64) {
        const double x = (double)px, y = (double)py;
        l1 = (float)__builtin_fma(q.A[2], x, __builtin_fma(q.B[2], y, q.C[2])) * q.inv_area;
        l2 = (float)__builtin_fma(q.A[0], x, __builtin_fma(q.B[0], y, q.C[0])) * q.inv_area;
    } else {
        Edges e;
        make_edges(t, e);
        const float inv_area = 1.0f / (float)t.area2;
        l1 = (float)edge_eval(e, 2, px, py) * inv_area;
        l2 = (float)edge_eval(e, 0, px, py) * inv_area;
    }
    const float l0 = (1.0f - l1) - l2;
    const float pw0 = l0 * t.iw[0], pw1 = l1 * t.iw[1], pw2 = l2 * t.iw[2];
    const float rr = 1.0f / ((pw0 + pw1) + pw2);
    const float b0 = pw0 * rr, b1 = pw1 * rr, b2 = pw2 * rr;
    if (q.flags & RASTER_UNIT_BARY) {
        // an uncut source triangle: the rows of t.bary are unit vectors, the product below is (b0, b1, b2) itself -- in source order when set-up
        // exchanged two vertices.  Taken HERE, for every consumer (k_resolve, k_material_vis, the integer-path records), so that the G-buffer
        // path and the visibility-plane path agree bit for bit also where the product would not return its operand: b * 1 + b' * 0 + b'' * 0
        // is NaN once a b is infinite or NaN (a degenerate 1/w sum), and -0 for a negative zero.
        const bool swapped = (q.flags & RASTER_SWAPPED) != 0u;
        B[0] = b0; B[1] = swapped ? b2 : b1; B[2] = swapped ? b1 : b2;
        return;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) B[k] = (b0 * t.bary[0][k] + b1 * t.bary[1][k]) + b2 * t.bary[2][k];
}
__device__ __forceinline__ float interpolate_attr(const float B[3], const float *A0, const float *A1, const float *A2, int k) {
#pragma clang fp contract(off)
    return (B[0] * A0[k] + B[1] * A1[k]) + B[2] * A2[k];
}

}  // namespace arctic
