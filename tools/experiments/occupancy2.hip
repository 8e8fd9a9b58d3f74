// What else besides the VGPR count decides how many waves a SIMD holds?  occupancy.hip's kernels (VGPRs only) reach the documented 512 / N;
// the shading kernel (72 VGPRs, 1 KiB of LDS per workgroup, ~97 SGPRs) was seen in 6 wave slots, not 7 (profiles/r3_tile_trace_*).
// Same experiment with LDS and SGPR use added.    hipcc -O3 --offload-arch=gfx950 tools/experiments/occupancy2.hip -o /tmp/occupancy2 && /tmp/occupancy2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
template <int NV, int LDS, int NS> __global__ __launch_bounds__(256) void k(unsigned *out, int spin) {
    __shared__ unsigned lds[LDS ? LDS / 4 : 1];
    unsigned x = threadIdx.x;
    if (LDS) lds[threadIdx.x % (LDS / 4)] = x;
    if (NV == 64) asm volatile("v_mov_b32 v63, %0" :: "v"(x) : "v63");
    if (NV == 72) asm volatile("v_mov_b32 v71, %0" :: "v"(x) : "v71");
    if (NV == 56) asm volatile("v_mov_b32 v55, %0" :: "v"(x) : "v55");
    if (NS == 96) asm volatile("s_mov_b32 s95, 0" ::: "s95");
    if (NS == 100) asm volatile("s_mov_b32 s99, 0" ::: "s99");
    if (NS == 80) asm volatile("s_mov_b32 s79, 0" ::: "s79");
    if (NS == 82) asm volatile("s_mov_b32 s81, 0" ::: "s81");
    if (NS == 84) asm volatile("s_mov_b32 s83, 0" ::: "s83");
    if (NS == 86) asm volatile("s_mov_b32 s85, 0" ::: "s85");
    if (NS == 88) asm volatile("s_mov_b32 s87, 0" ::: "s87");
    if (NS == 90) asm volatile("s_mov_b32 s89, 0" ::: "s89");
    if (NS == 92) asm volatile("s_mov_b32 s91, 0" ::: "s91");
    if (NS == 94) asm volatile("s_mov_b32 s93, 0" ::: "s93");
    if (NS == 72) asm volatile("s_mov_b32 s71, 0" ::: "s71");
    if (NS == 74) asm volatile("s_mov_b32 s73, 0" ::: "s73");
    if (NS == 76) asm volatile("s_mov_b32 s75, 0" ::: "s75");
    if (NS == 78) asm volatile("s_mov_b32 s77, 0" ::: "s77");
    if (NS == 64) asm volatile("s_mov_b32 s63, 0" ::: "s63");
    if (NS == 70) asm volatile("s_mov_b32 s69, 0" ::: "s69");
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin) { asm volatile("s_sleep 8"); }
    const unsigned hw = __builtin_amdgcn_s_getreg(4 | (31 << 11)), xcc = __builtin_amdgcn_s_getreg(20 | (31 << 11));
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw + (LDS ? lds[0] * 0 : 0); out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc; }
}
template <int NV, int LDS, int NS> void run(unsigned *d, std::vector<unsigned> &h) {
    const int blocks = 256 * 10;
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k<NV, LDS, NS>, 256, 0);
    hipFuncAttributes fa; hipFuncGetAttributes(&fa, (const void *)k<NV, LDS, NS>);
    k<NV, LDS, NS><<<blocks, 256>>>(d, 2000);
    hipMemcpy(h.data(), d, blocks * 4 * 8, hipMemcpyDeviceToHost);
    int maxslot = 0;
    for (int i = 0; i < blocks * 4; ++i) maxslot = std::max(maxslot, (int)(h[2 * i] & 15));
    printf("%3d VGPRs (%3d in the code object), %5d B LDS per workgroup, SGPR s%d touched: runtime predicts %d workgroups per CU; wave slot ids seen 0..%d = %d waves per SIMD\n", NV, fa.numRegs, LDS, NS - 1, occ, maxslot, maxslot + 1);
}
int main() {
    unsigned *d; hipMalloc(&d, 256 * 10 * 4 * 8);
    std::vector<unsigned> h(256 * 10 * 4 * 2);
    run<72, 0, 0>(d, h); run<72, 1024, 0>(d, h); run<72, 1024, 80>(d, h); run<72, 1024, 82>(d, h); run<72, 1024, 84>(d, h); run<72, 1024, 86>(d, h); run<72, 1024, 88>(d, h);
    run<72, 1024, 90>(d, h); run<72, 1024, 92>(d, h); run<72, 1024, 94>(d, h); run<72, 1024, 96>(d, h); run<72, 1024, 100>(d, h);
    run<64, 0, 0>(d, h); run<64, 1024, 64>(d, h); run<64, 1024, 70>(d, h); run<64, 1024, 72>(d, h); run<64, 1024, 74>(d, h); run<64, 1024, 76>(d, h); run<64, 1024, 78>(d, h);
    run<64, 1024, 80>(d, h); run<64, 1024, 88>(d, h); run<64, 1024, 96>(d, h);
    run<56, 1024, 96>(d, h); run<56, 1024, 80>(d, h); run<56, 1024, 72>(d, h);
    return 0;
}
