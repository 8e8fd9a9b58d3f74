// loop_rate.hip -- the packed light loop of shade.hip IN ISOLATION: issue cycles per pair trip with W waves per SIMD (round 5).
// The shading kernels' own functions (accumulate_pair, load_light_pair, wait_and_sub: shade.hip is included as a whole) in a kernel that does nothing else:
// per-pixel constants from a seed, N_TRIPS trips over a small light table in the scalar cache, the nine sums written out at the end.
// If the loop runs at its census price here (48 packed x ~4.3 + 6 transcendentals x 8 = ~254 cycles) and the pass does not, the pass loses time AROUND the loop.
//   hipcc -O3 -std=c++17 -ffp-contract=off -mllvm -disable-machine-licm --offload-arch=gfx950 -I arctic-renderer_amd/csrc tools/experiments/loop_rate.hip -o build_tmp/loop_rate
#include "../../arctic-renderer_amd/csrc/shade.hip"
#include <cstdio>
#include <vector>
using namespace arctic;
namespace arctic { namespace {
constexpr int N_TRIPS = 2048, N_PAIRS = 32;
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_loop(const float4 *pairs, float *out, float seed) {
    const uint32_t lane = wave_lane();
    LoopPix px;
    TailPix tp;
    const float u = seed + (float)lane * 0.01f;
    make_pix(normalize(mk(0.1f + u, 1.0f, 0.2f)), normalize(mk(0.3f, 0.8f + u, 0.5f)), mk(u, 0.5f * u, 1.0f - u), mk(0.5f, 0.4f, 0.3f), 0.2f, 0.4f + 0.1f * u, px, tp);
    const PackedPix pk = pack_pix(px);
    Sums2 S;
    for (int k = 0; k < 3; ++k) { S.a[k] = (v2){0.0f, 0.0f}; S.b[k] = (v2){0.0f, 0.0f}; S.c[k] = (v2){0.0f, 0.0f}; }
    const char *lp = reinterpret_cast<const char *>(pairs);
    f4v A0, B0, C0, A1, B1, C1;
    v2 dx, dy, dz;
    load_light_pair(lp, A0, B0, C0);
    for (int t = 0; t < N_TRIPS; t += 2) {
        wait_and_sub((v2){A0.x, A0.y}, (v2){A0.z, A0.w}, (v2){B0.x, B0.y}, pk.w_xy, pk.wz_a2, dx, dy, dz);
        load_light_pair(lp + 48 * ((t + 1) % N_PAIRS), A1, B1, C1);
        accumulate_pair(pk, dx, dy, dz, (v2){B0.z, B0.w}, (v2){C0.x, C0.y}, (v2){C0.z, C0.w}, S);
        wait_and_sub((v2){A1.x, A1.y}, (v2){A1.z, A1.w}, (v2){B1.x, B1.y}, pk.w_xy, pk.wz_a2, dx, dy, dz);
        load_light_pair(lp + 48 * ((t + 2) % N_PAIRS), A0, B0, C0);
        accumulate_pair(pk, dx, dy, dz, (v2){B1.z, B1.w}, (v2){C1.x, C1.y}, (v2){C1.z, C1.w}, S);
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    float acc = 0.0f;
    for (int k = 0; k < 3; ++k) acc += S.a[k].x + S.a[k].y + S.b[k].x + S.b[k].y + S.c[k].x + S.c[k].y;
    if (acc == 12345.678f) out[0] = acc + tp.num;
}
} }
int main() {
    std::vector<float4> h(3 * N_PAIRS);
    for (int p = 0; p < N_PAIRS; ++p) {
        h[3 * p] = make_float4(1.0f + p, 2.0f + p, 3.0f, 4.0f + 0.5f * p);
        h[3 * p + 1] = make_float4(-2.0f + 0.3f * p, 1.5f, 3.0f, 2.0f);
        h[3 * p + 2] = make_float4(1.0f, 4.0f, 2.0f, 0.5f);
    }
    float4 *d; float *out;
    hipMalloc(&d, h.size() * 16); hipMalloc(&out, 4);
    hipMemcpy(d, h.data(), h.size() * 16, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("packed light loop alone: cycles per pair trip per SIMD (at the nominal 2.4 GHz; the chip's clock under this load is lower: compare W = 1 with the instruction table)\n");
    for (int W : {1, 2, 4, 6, 7, 8}) {
        const int blocks = 256 * W;
        k_loop<<<blocks, 256>>>(d, out, 0.25f); hipDeviceSynchronize();
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0); k_loop<<<blocks, 256>>>(d, out, 0.25f); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
        }
        // a long run for the clock to settle: 40 launches back to back
        hipEventRecord(e0); for (int rep = 0; rep < 40; ++rep) k_loop<<<blocks, 256>>>(d, out, 0.25f); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms40 = 0; hipEventElapsedTime(&ms40, e0, e1);
        printf("W = %d waves per SIMD: %.1f cycles per trip and SIMD (best of 5 launches), %.1f sustained over 40 launches back to back (%.2f ms each)\n", W,
               best * 1e-3 * 2.4e9 / ((double)N_TRIPS * W), ms40 / 40 * 1e-3 * 2.4e9 / ((double)N_TRIPS * W), ms40 / 40);
    }
    return 0;
}
