// div_near_one.hip -- EXHAUSTIVE check of a short exact division x / w for divisors within a few ulps of 1.0 (round 5).
// Why: calculate_shadow divides the interpolated light-space position by its w (forward.hlsl:70); the reference's sun is orthographic
// (scene.cpp:61-70), so w is 1 at every vertex and (b0 + b1) + b2 = 1 -+ a few ulps after interpolation -- in config 3 NOT ONE 8x8 tile has w == 1 in
// every pixel, and every tile paid three IEEE divisions (v_div_scale / v_rcp / 7 x fma / v_div_fmas / v_div_fixup: 33 VALU + 3 transcendentals).
// The quotient feeds floor() and the 25 PCF compares, so it must be the IEEE quotient bit for bit.  Candidates, y = v_rcp_f32(w) refined once:
//   A: q0 = x y; r = fma(-w, q0, x); q = fma(r, y, q0)                       (one correction)
//   B: A, then r = fma(-w, q, x); q = fma(r, y, q)                            (two corrections: the compiler's sequence without its scaling / fix-up)
//   A0: A with the raw v_rcp_f32(w);  C1 / C2: y = 2 - w (no transcendental at all: 1 / (1 + d) = 1 - d + ...), one / two corrections
// For every w = 1.0 +- k ulp-steps, k = 0..6 (13 divisors), ALL 2^32 bit patterns of x are compared with the compiler's IEEE x / w.
//   hipcc -O3 -ffp-contract=off --offload-arch=gfx950 tools/experiments/div_near_one.hip -o build_tmp/div_near_one && build_tmp/div_near_one
#include <hip/hip_runtime.h>
#include "../../arctic-renderer_amd/csrc/shadow_coords.h"   // quotient_near_one, near_one_divisible: the functions the kernels use
#include <cstdio>
#include <cstdint>
#include <cstring>

__device__ __forceinline__ float recip_refined(float w) {
    float y = __builtin_amdgcn_rcpf(w);
    return __builtin_fmaf(__builtin_fmaf(-w, y, 1.0f), y, y);
}
__device__ __forceinline__ float div_a(float x, float w, float y) { return arctic::quotient_near_one(x, w, y); }
__device__ __forceinline__ float div_b(float x, float w, float y) {
    const float q1 = div_a(x, w, y);
    return __builtin_fmaf(__builtin_fmaf(-w, q1, x), y, q1);
}
__device__ __forceinline__ bool same(float a, float b) {
    return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b);
}
// variant v = 0..4: A, B, A0, C1, C2.  out[v]: mismatches over x = 0 or 2^-100 < |x| < 2^100 (-0 against +0 not counted: the quotient only feeds
// additions and compares); out[8 + v]: mismatches outside that range (reported apart: the kernel guards it); out[16 + v]: first example.  OLD: out[2..3]: over x whose quotient is below 2^-100 in magnitude or above 2^100 (reported apart); out[4..7]: first examples
__global__ __launch_bounds__(256) void k_check(uint32_t w_bits, unsigned long long *out) {
    const float w = __uint_as_float(w_bits);
    const float y = recip_refined(w), y0 = __builtin_amdgcn_rcpf(w), yc = 2.0f - w;
    unsigned long long bad[5] = {0, 0, 0, 0, 0}, edge[5] = {0, 0, 0, 0, 0};
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < (1ull << 32); i += (unsigned long long)gridDim.x * 256) {
        const float x = __uint_as_float((uint32_t)i);
        volatile float wv = w;             // (the compiler must not fold the division)
        const float ref = x / wv;
        const float c[5] = {div_a(x, w, y), div_b(x, w, y), div_a(x, w, y0), div_a(x, w, yc), div_b(x, w, yc)};
        const bool in_range = arctic::near_one_divisible(x, x, x, w);   // the kernels' own guard (x = 0 is outside it)
#pragma unroll
        for (int v = 0; v < 5; ++v) {
            const bool ok = same(c[v], ref) || (c[v] == 0.0f && ref == 0.0f);
            if (!ok) { if (in_range) { ++bad[v]; atomicMin(&out[16 + v], i); } else ++edge[v]; }
        }
    }
    for (int v = 0; v < 5; ++v) { atomicAdd(&out[v], bad[v]); atomicAdd(&out[8 + v], edge[v]); }
}

int main() {
    unsigned long long *d, h[24];
    hipMalloc(&d, sizeof h);
    int worst = 0;
    const char *names[5] = {"A (rcp refined, 1 correction)", "B (rcp refined, 2 corrections)", "A0 (raw rcp, 1 correction)", "C1 (y = 2 - w, 1 correction)", "C2 (y = 2 - w, 2 corrections)"};
    for (int k = -6; k <= 6; ++k) {
        const uint32_t w_bits = 0x3F800000u + (uint32_t)k;
        float w; memcpy(&w, &w_bits, 4);
        unsigned long long init[24] = {};
        for (int v = 0; v < 8; ++v) init[16 + v] = ~0ull;
        hipMemcpy(d, init, sizeof h, hipMemcpyHostToDevice);
        k_check<<<4096, 256>>>(w_bits, d);
        hipDeviceSynchronize();
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("w = 1%+d ulp-steps (0x%08x = %.9g):", k, w_bits, w);
        for (int v = 0; v < 5; ++v) {
            printf("  %s: %llu (+%llu out of range)", names[v], h[v], h[8 + v]);
            if (h[v]) printf(" first x = 0x%08llx", h[16 + v]);
        }
        printf("\n"); fflush(stdout);
        if (h[2]) worst = 1;   // A0 is what the kernels run
    }
    return worst;
}
