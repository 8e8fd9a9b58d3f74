// recip_exact.hip -- EXHAUSTIVE check of a short exact reciprocal 1 / x (round 5): the perspective correction of the barycentrics divides by the
// interpolated 1/w (edges.h source_barycentrics: rr = 1 / ((pw0 + pw1) + pw2)), once per covered pixel in k_resolve and k_material_vis, as a full IEEE
// division (v_div_scale x 2, v_rcp, 7 x fma, v_div_fmas, v_div_fixup); the quotient feeds attributes that must match the oracle bit for bit.
//   R1: y0 = v_rcp_f32(x); y = fma(fma(-x, y0, 1), y0, y0)          R2: R1, then y = fma(fma(-x, y, 1), y, y)
// against the compiler's 1.0f / x for ALL 2^32 bit patterns of x; reported apart: x outside 2^-100 < |x| < 2^100 (a guard the kernels would carry).
//   hipcc -O3 -ffp-contract=off --offload-arch=gfx950 tools/experiments/recip_exact.hip -o build_tmp/recip_exact && build_tmp/recip_exact
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ bool same(float a, float b) { return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b); }
__global__ __launch_bounds__(256) void k_check(unsigned long long *out) {
    unsigned long long bad[2] = {0, 0}, edge[2] = {0, 0};
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < (1ull << 32); i += (unsigned long long)gridDim.x * 256) {
        const float x = __uint_as_float((uint32_t)i);
        volatile float one = 1.0f;
        const float ref = one / x;
        const float y0 = __builtin_amdgcn_rcpf(x);
        const float y1 = __builtin_fmaf(__builtin_fmaf(-x, y0, 1.0f), y0, y0);
        const float y2 = __builtin_fmaf(__builtin_fmaf(-x, y1, 1.0f), y1, y1);
        const float ax = fabsf(x);
        const bool in_range = ax > 0x1p-100f && ax < 0x1p100f;
        const float c[2] = {y1, y2};
        for (int v = 0; v < 2; ++v)
            if (!same(c[v], ref)) { if (in_range) { ++bad[v]; atomicMin(&out[4 + v], i); } else ++edge[v]; }
    }
    for (int v = 0; v < 2; ++v) { atomicAdd(&out[v], bad[v]); atomicAdd(&out[2 + v], edge[v]); }
}
int main() {
    unsigned long long *d, h[6] = {0, 0, 0, 0, ~0ull, ~0ull};
    hipMalloc(&d, sizeof h);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    k_check<<<4096, 256>>>(d);
    hipDeviceSynchronize();
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("1 / x, all 2^32 x: one refinement %llu mismatches in range (+%llu outside), first x = 0x%08llx; two refinements %llu (+%llu), first x = 0x%08llx\n",
           h[0], h[2], h[4], h[1], h[3], h[5]);
    return 0;
}
