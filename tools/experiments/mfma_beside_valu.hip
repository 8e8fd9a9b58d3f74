// mfma_beside_valu.hip -- does a small fp32 MFMA issue in the shadow of packed-fp32 VALU work on gfx950, and what does it compute?
//   hipcc -O3 --offload-arch=gfx950 tools/experiments/mfma_beside_valu.hip -o /tmp/mfma_beside_valu && /tmp/mfma_beside_valu
// Part 1: layout of v_mfma_f32_4x4x1_16b_f32 (16 blocks of D[4x4] += A[4x1] B[1x4]): which lane / register holds D[block][i][j].
// Part 2: time per trip of (a) 48 v_pk_fma_f32, (b) 39 of them, (c) 39 + 6 MFMAs interleaved, (d) the 6 MFMAs alone,
//         W = 1..8 waves per SIMD, all SIMDs busy -- the light loop's 9 colour accumulations as 6 rank-1 MFMA updates.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float v2 __attribute__((ext_vector_type(2)));
constexpr int N_IT = 4096;

__global__ void k_layout(float *out) {
    const int l = threadIdx.x;
    f4 c = {0, 0, 0, 0};
    // A[block][i] = 100 block + 10 i + 1 (lane 4 block + i), B[block][j] = 1000 + j (lane 4 block + j): D[block][i][j] = A B
    c = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(100 * (l / 4) + 10 * (l % 4) + 1), (float)(1000 + l % 4), c, 0, 0, 0);
    out[l * 4 + 0] = c.x; out[l * 4 + 1] = c.y; out[l * 4 + 2] = c.z; out[l * 4 + 3] = c.w;
}

#define PK(r) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(r) : "v"(ps));
#define MF(acc, a) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
#define PK13 PK(p0) PK(p1) PK(p2) PK(p3) PK(p4) PK(p5) PK(p6) PK(p7) PK(p8) PK(p9) PK(p10) PK(p11) PK(p12)
#define PK3 PK(p13) PK(p14) PK(p15)
#define DECL                                                                                                  \
    v2 ps = {seed, seed + 1.0f};                                                                              \
    v2 p0 = ps, p1 = ps * 2.f, p2 = ps * 3.f, p3 = ps, p4 = ps, p5 = ps, p6 = ps, p7 = ps, p8 = ps, p9 = ps, p10 = ps, p11 = ps, p12 = ps, p13 = ps, p14 = ps, p15 = ps; \
    f4 a0 = {seed, 0, 0, 0}, a1 = a0, a2 = a0;                                                                \
    float b = seed;
#define FIN                                                                                                   \
    float s = p0.x + p1.x + p2.x + p3.x + p4.x + p5.x + p6.x + p7.x + p8.x + p9.x + p10.x + p11.x + p12.x + p13.x + p14.x + p15.x + a0.x + a1.y + a2.z; \
    if (s == 12345.f) out[0] = s;

__global__ __launch_bounds__(256) void k_pk48(float *out, float seed) {
    DECL
    for (int it = 0; it < N_IT; ++it) { PK13 PK3 PK13 PK3 PK13 PK3 }
    FIN
}
__global__ __launch_bounds__(256) void k_pk39(float *out, float seed) {
    DECL
    for (int it = 0; it < N_IT; ++it) { PK13 PK13 PK13 }
    FIN
}
__global__ __launch_bounds__(256) void k_pk39_mfma6(float *out, float seed) {
    DECL
    for (int it = 0; it < N_IT; ++it) {
        PK13 MF(a0, p0.x) MF(a1, p1.x) MF(a2, p2.x) PK13 MF(a0, p0.y) MF(a1, p1.y) MF(a2, p2.y) PK13
    }
    FIN
}
__global__ __launch_bounds__(256) void k_pk39_mfma6_spread(float *out, float seed) {
    DECL
    for (int it = 0; it < N_IT; ++it) {
        PK(p0) PK(p1) PK(p2) PK(p3) PK(p4) PK(p5) MF(a0, p0.x) PK(p6) PK(p7) PK(p8) PK(p9) PK(p10) PK(p11) MF(a1, p1.x) PK(p12)
        PK(p0) PK(p1) PK(p2) PK(p3) PK(p4) MF(a2, p2.x) PK(p5) PK(p6) PK(p7) PK(p8) PK(p9) PK(p10) MF(a0, p0.y) PK(p11) PK(p12)
        PK(p0) PK(p1) PK(p2) PK(p3) PK(p4) PK(p5) MF(a1, p1.y) PK(p6) PK(p7) PK(p8) PK(p9) PK(p10) PK(p11) MF(a2, p2.y) PK(p12)
    }
    FIN
}
__global__ __launch_bounds__(256) void k_mfma6(float *out, float seed) {
    DECL
    for (int it = 0; it < N_IT; ++it) { MF(a0, p0.x) MF(a1, p1.x) MF(a2, p2.x) MF(a0, p0.y) MF(a1, p1.y) MF(a2, p2.y) }
    FIN
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <class K> static double time_kernel(K kern, int waves_per_simd, float *out) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 256 * waves_per_simd;   // 256 CUs x waves_per_simd blocks of 4 waves
    kern<<<blocks, 256>>>(out, 1.0f);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        (void)hipEventRecord(e0);
        kern<<<blocks, 256>>>(out, 1.0f);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    return best * 1e6 / N_IT;   // ns per trip
}

int main() {
    float *out;
    CHECK(hipMalloc(&out, 64 * 4 * sizeof(float)));
    k_layout<<<1, 64>>>(out);
    std::vector<float> h(256);
    CHECK(hipMemcpy(h.data(), out, 256 * 4, hipMemcpyDeviceToHost));
    printf("layout of v_mfma_f32_4x4x1_16b_f32 with A[block][i] = 100 block + 10 i + 1 in lane 4 block + i, B[block][j] = 1000 + j in lane 4 block + j:\n");
    for (int l : {0, 1, 2, 3, 4, 5, 62, 63}) printf("  lane %2d: d = %.0f %.0f %.0f %.0f\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    bool as_assumed = true;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) as_assumed = as_assumed && h[l * 4 + r] == (float)(100 * (l / 4) + 10 * r + 1) * (float)(1000 + l % 4);
    printf("  D[block][i][j] lives in lane 4 block + j, register i: %s\n", as_assumed ? "yes" : "NO");
    printf("ns per trip (all SIMDs busy):      W=1      W=2      W=4      W=7      W=8\n");
    const int Ws[5] = {1, 2, 4, 7, 8};
    printf("48 v_pk_fma_f32                 "); for (int w : Ws) printf(" %8.2f", time_kernel(k_pk48, w, out) / w); printf("   (per wave on the SIMD)\n");
    printf("39 v_pk_fma_f32                 "); for (int w : Ws) printf(" %8.2f", time_kernel(k_pk39, w, out) / w); printf("\n");
    printf("39 v_pk_fma_f32 + 6 MFMA, 3 + 3 "); for (int w : Ws) printf(" %8.2f", time_kernel(k_pk39_mfma6, w, out) / w); printf("\n");
    printf("39 v_pk_fma_f32 + 6 MFMA spread "); for (int w : Ws) printf(" %8.2f", time_kernel(k_pk39_mfma6_spread, w, out) / w); printf("\n");
    printf("6 MFMA 4x4x1 alone              "); for (int w : Ws) printf(" %8.2f", time_kernel(k_mfma6, w, out) / w); printf("\n");
    return 0;
}
