// How many waves does a SIMD of this box really hold as a function of a kernel's VGPR count?  Kernels that touch v(N-1) spin for a
// while; every wave records its HW_ID; the answer is the number of distinct wave slots seen per SIMD (and what the runtime predicts).
//   hipcc -O3 --offload-arch=gfx950 tools/experiments/occupancy.hip -o /tmp/occupancy && /tmp/occupancy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>
template <int N> __global__ __launch_bounds__(256) void k(unsigned *out, int spin) {
    unsigned x = threadIdx.x;
    if (N == 64) asm volatile("v_mov_b32 v63, %0" :: "v"(x) : "v63");
    if (N == 72) asm volatile("v_mov_b32 v71, %0" :: "v"(x) : "v71");
    if (N == 80) asm volatile("v_mov_b32 v79, %0" :: "v"(x) : "v79");
    if (N == 88) asm volatile("v_mov_b32 v87, %0" :: "v"(x) : "v87");
    if (N == 96) asm volatile("v_mov_b32 v95, %0" :: "v"(x) : "v95");
    if (N == 104) asm volatile("v_mov_b32 v103, %0" :: "v"(x) : "v103");
    if (N == 128) asm volatile("v_mov_b32 v127, %0" :: "v"(x) : "v127");
    if (N == 56) asm volatile("v_mov_b32 v55, %0" :: "v"(x) : "v55");
    if (N == 48) asm volatile("v_mov_b32 v47, %0" :: "v"(x) : "v47");
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin) { asm volatile("s_sleep 8"); }
    const unsigned hw = __builtin_amdgcn_s_getreg(4 | (31 << 11)), xcc = __builtin_amdgcn_s_getreg(20 | (31 << 11));
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw; out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc; }
}
template <int N> void run(unsigned *d, std::vector<unsigned> &h) {
    const int blocks = 256 * 10;
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k<N>, 256, 0);
    hipFuncAttributes fa; hipFuncGetAttributes(&fa, (const void *)k<N>);
    k<N><<<blocks, 256>>>(d, 2000);   // 20 us each: ten rounds' worth of blocks queue up, the first fill every slot
    hipMemcpy(h.data(), d, blocks * 4 * 8, hipMemcpyDeviceToHost);
    std::set<unsigned> slots;
    int maxslot = 0;
    for (int i = 0; i < blocks * 4; ++i) { const unsigned hw = h[2 * i]; maxslot = std::max(maxslot, (int)(hw & 15)); }
    printf("%3d VGPRs asked, %3d in the code object: runtime predicts %d workgroups (of 4 waves) per CU = %d waves per SIMD; wave slot ids seen 0..%d\n", N, fa.numRegs, occ, occ, maxslot);
}
int main() {
    unsigned *d; hipMalloc(&d, 256 * 10 * 4 * 8);
    std::vector<unsigned> h(256 * 10 * 4 * 2);
    run<48>(d, h); run<56>(d, h); run<64>(d, h); run<72>(d, h); run<80>(d, h); run<88>(d, h); run<96>(d, h); run<104>(d, h); run<128>(d, h);
    return 0;
}
